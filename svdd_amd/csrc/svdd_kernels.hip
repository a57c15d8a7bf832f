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
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
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
// exp of a SMALL non-positive argument (-2^-7 < x <= 0), the same contract (fp64 value within an ulp of double, rounded once to
// fp32): degree-6 Taylor in fp64 (the next term is < 2^-61): 7 fp64 FMAs instead of ocml's range reduction + degree-11 kernel.
// K2's exact softmax path evaluates exp(s - max) of scores that lie ~1e-7 apart — the regime random-init value nets produce.
__device__ __forceinline__ float expf_cr_small(float x) {
  const double d = (double)x;
  double p = 1.0 / 720.0;
  p = __builtin_fma(p, d, 1.0 / 120.0);
  p = __builtin_fma(p, d, 1.0 / 24.0);
  p = __builtin_fma(p, d, 1.0 / 6.0);
  p = __builtin_fma(p, d, 0.5);
  p = __builtin_fma(p, d, 1.0);
  p = __builtin_fma(p, d, 1.0);
  // The series and ocml's exp are each within an ulp of double of the true value, so they round to the same fp32 unless the
  // 29 discarded bits sit within a few double-ulps of the fp32 rounding midpoint (2^-26 of the arguments): there, defer to
  // ocml itself — expf_cr_small(x) == expf_cr(x) for every x, not just with overwhelming probability.
  const long long low = __double_as_longlong(p) & 0x1FFFFFFFll;
  if (low - 0x0FFFFFF8ll <= 16ll && low >= 0x0FFFFFF8ll) return expf_cr(x);
  return (float)p;
}
__device__ __forceinline__ float expf_cr_nonpos(float x) { return x > -0.0078125f ? expf_cr_small(x) : expf_cr(x); }

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

// One Philox block (128 bits) feeds the 5 uniforms of a draw (5 x 24 = 120 bits): categories 0..3
// take the top 24 bits of the four words, MASK takes the low bytes of words 0..2.
__device__ __forceinline__ void philox_uniform5(uint64_t seed, uint64_t pos, uint32_t step, uint32_t m,
                                                float (&u)[V]) {
  uint32_t c[4] = {(uint32_t)pos, (uint32_t)(pos >> 32), (step << 16) | m, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  u[0] = u24(c[0]); u[1] = u24(c[1]); u[2] = u24(c[2]); u[3] = u24(c[3]);
  const uint32_t low = (c[0] & 0xFFu) | ((c[1] & 0xFFu) << 8) | ((c[2] & 0xFFu) << 16);
  u[4] = (float)low * (1.0f / 16777216.0f);
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
// The exact semantics (oracle) evaluate exp/log correctly rounded, which on the GPU means fp64
// (ocml) — ~100 quarter-rate instructions per transcendental, 5 per draw: compute-bound far below
// the HBM roofline. The only OUTPUT that depends on them in the hot loop is an argmax, so K1 uses an
// exact-arithmetic *filter*: a fast fp32 path (v_exp_f32 / v_log_f32 / v_rcp_f32, a few ulps off)
// decides every draw whose winner leads the runner-up by more than a proven error margin; the rare
// ambiguous lane (~1e-5 of draws) re-evaluates in the exact arithmetic. Results are identical to
// the exact path by construction (DESIGN.md "K1 filter"; bound checked exhaustively over all 2^24
// possible uniforms by svdd_selftest_fastmath and A/B-tested against the forced-exact path).
//
// Block = 256 threads = 4 waves over one tile of 64 consecutive (b,l) positions; every wave rebuilds
// the tile's (cheap, fast-path) q in registers — no LDS, no barrier — and the waves split the M
// candidates. Stores: cand 1 B/lane (64 B per wave-store), one-hot float4/lane (1 KiB per wave-store).
struct ProposeArgs {
  const float* logits; const uint8_t* x; float dm, mcs; int B, L, M, layout;
  int rng_kind; uint32_t step; const float* uniforms; uint64_t seed, row_offset;
  uint8_t* cand; float* onehot; float* q_xs; int force_exact; int msplit; int ulayout;
  unsigned long long* stats;     // optional device counters {masked draws, draws sent to the exact path} (svdd_k1_stats)
  int tok_rows;                  // candidates per chunk of the LDS token table = min(64, candidates of a unit)
  int u_rows;                    // REPLAY: rows of the WHOLE batch the uniform blocks hold (>= B; this shard's row 0 = row_offset)
};

constexpr float LOG2E_HI = 1.44269502162933349609375f;        // fl32(log2 e)
constexpr float LOG2E_LO = 1.925963033500011e-08f;             // log2 e - LOG2E_HI
constexpr float LN2 = 0.693147182464599609375f;

// exp(x), relative error <= ~2^-21 for -80 < x <= 0 (hardware exp2 is 1 ulp; the product rounding
// of x*log2e is compensated with an fma residual).
__device__ __forceinline__ float exp_fast(float x) {
  const float t = x * LOG2E_HI;
  const float r = __fmaf_rn(x, LOG2E_HI, -t) + x * LOG2E_LO;   // x*log2e - t
  const float e = __builtin_amdgcn_exp2f(t);
  return __fmaf_rn(e, r * LN2, e);
}
__device__ __forceinline__ float log_fast(float a) { return __builtin_amdgcn_logf(a) * LN2; }   // rel err <= ~2^-22

// exact q for one position (oracle arithmetic)
__device__ __forceinline__ void q_exact(const float (&z)[V], int xt, float dm, float mcs, float (&q)[V]) {
  if (xt != MASK) {               // exp(0)*dm and exp(-1e6)*dm: constants, no transcendental needed
#pragma unroll
    for (int v = 0; v < V; ++v) q[v] = (v == xt) ? dm : 0.0f;
  } else {
    float lp[V];
    subs_logp_1(z, xt, lp);
#pragma unroll
    for (int v = 0; v < V; ++v) q[v] = expf_cr(lp[v]) * dm;     // diffusion_gosai.py:1194
  }
  q[MASK] = mcs;                                                 // :1196
}

// Work decomposition: a *unit* = (tile of 64 consecutive (b,l) positions, candidate group s of
// `msplit`): the wave that owns a unit rebuilds the tile's cheap fast-path q in registers (no LDS, no
// barrier) and draws candidates m = s, s + msplit, ... Waves take units in a grid-stride loop, so a
// large launch runs as persistent blocks (as many as the chip holds at once) instead of one short-lived wave per unit (wave
// dispatch was the bottleneck of the one-wave-per-unit version). msplit is chosen by the host: 1 when
// there are enough tiles to fill the chip, up to 4 for small batches where latency dominates.
template <bool REPLAY, bool QGIVEN>
__global__ __launch_bounds__(256) void propose_kernel(ProposeArgs a) {
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t N = (uint32_t)a.B * (uint32_t)a.L;            // host guarantees B*L*5*M < 2^31-ish for 32-bit tiles
  const uint32_t ntiles = (N + WAVE - 1) / WAVE;
  const uint32_t nunits = ntiles * (uint32_t)a.msplit;
  const uint32_t wave_global = blockIdx.x * 4u + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4u;
  uint32_t n_draws = 0, n_exact = 0;
  // per wave, by rank of the masked position: q~[5], margin, owner lane, z[5], then the position's RNG key (Philox: the
  // 64-bit global position, lo / hi ; replay: b, l) so that a draw needs no division or 64-bit multiply of its own
  __shared__ float prm[4][14 * WAVE];
  __shared__ float4 oh_lut[V];                 // one-hot rows by token (MASK: zeros): one ds_read_b128 instead of 8 VALU ops
  // per wave: drawn tokens [candidate of the chunk][lane of the position]; sized by the host for min(64, candidates of a
  // unit) rows, so that at M = 10 a block holds 15 KB of LDS instead of 28 and the CU's resident waves are limited by
  // registers (7 per SIMD), not LDS (5): more waves to overlap one wave's draws with another's stores
  extern __shared__ __attribute__((aligned(16))) uint8_t tok_dyn[];
  const int wv = threadIdx.x >> 6;
  const int CR = a.tok_rows;
  uint8_t* tokw = tok_dyn + (size_t)wv * CR * WAVE;

  if (threadIdx.x < V)
    oh_lut[threadIdx.x] = float4{threadIdx.x == 0 ? 1.0f : 0.0f, threadIdx.x == 1 ? 1.0f : 0.0f, threadIdx.x == 2 ? 1.0f : 0.0f,
                                 threadIdx.x == 3 ? 1.0f : 0.0f};
  __syncthreads();

  for (uint32_t unit = wave_global; unit < nunits; unit += nwaves) {
    const uint32_t tile = unit / (uint32_t)a.msplit;
    const int s0 = (int)(unit - tile * (uint32_t)a.msplit);
    const uint32_t n = tile * WAVE + lane;
    const bool valid = n < N;                 // lanes past the end of the batch stay in the loop: they take draws too
    const uint32_t nn = valid ? n : N - 1;
    // (b, l) of the lane's position: ONE wave-uniform division per tile, then a small per-lane remainder. Integer
    // multiplies and divides are quarter rate; the generic 64-bit index arithmetic of this prologue used to be ~50 of them.
    const uint32_t b0 = __builtin_amdgcn_readfirstlane((tile * WAVE) / (uint32_t)a.L);
    uint32_t rr = min(tile * WAVE + (uint32_t)lane, N - 1) - b0 * (uint32_t)a.L;      // < L + 64
    uint32_t b = b0;
    if ((uint32_t)a.L >= WAVE) { if (rr >= (uint32_t)a.L) { rr -= (uint32_t)a.L; ++b; } }
    else { const uint32_t qd = rr / (uint32_t)a.L; b += qd; rr -= qd * (uint32_t)a.L; }
    const uint32_t l = rr;

    // issue the token and logit loads together (the logits of an unmasked position are simply unused)
    const int xt = a.x[nn];
    float z[V];
    {
      const bool blv = a.layout == SVDD_LAYOUT_BLV;
      const float* zp = a.logits + (blv ? (uint64_t)nn * V : (uint64_t)b * (uint64_t)(V * a.L) + l);
      const uint32_t vs = blv ? 1u : (uint32_t)a.L;
#pragma unroll
      for (int v = 0; v < V; ++v) z[v] = zp[v * vs];
    }
    const bool masked = valid && xt == MASK;

    if (!QGIVEN && a.q_xs && s0 == 0 && valid) {       // per-step API only: q_xs is returned to the caller (:1228)
      float q[V];
      q_exact(z, xt, a.dm, a.mcs, q);
#pragma unroll
      for (int v = 0; v < V; ++v) a.q_xs[at(a.layout, b, l, v, a.L)] = q[v];
    }

    // fast-path q~ and the decision margin for this position
    float qf[V];
    float margin = 0.0f;                      // a draw is decided when second < best * margin
    bool fast_ok = false;
    if (QGIVEN) {                             // svdd_sample_categorical: `logits` already holds q (e.g. DPS-guided)
      fast_ok = !a.force_exact;
#pragma unroll
      for (int v = 0; v < V; ++v) { qf[v] = z[v]; fast_ok = fast_ok && (z[v] >= 0.0f) && (z[v] < 1e30f); }
      margin = 1.0f - 7.62939453125e-06f;     // 2^-17: only g~, rcp and the product carry error
    } else if (masked) {
      const float mx = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
      fast_ok = !a.force_exact && (fabsf(mx) < 60.0f) && (fabsf(z[MASK]) < 1e5f);   // also false on NaN
      if (fast_ok) {
        const float sum = (exp_fast(z[0] - mx) + exp_fast(z[1] - mx)) + (exp_fast(z[2] - mx) + exp_fast(z[3] - mx));
        const float lse = log_fast(sum) + mx;
#pragma unroll
        for (int v = 0; v < 4; ++v) qf[v] = exp_fast(z[v] - lse) * a.dm;
        qf[MASK] = a.mcs;
        // relative error budget of r~ = q~ * rcp(g~) against the exact quotient (DESIGN.md "K1 filter"):
        //   q~: 2^-19 + 2^-21 |lse| ; g~: 2^-21 (selftest: 2^-22.5) ; rcp+mul: 2^-22 ; x2 for the pair,
        //   + 2^-21 for the exact side's own rounding, x2 safety  =>  2^-16 + 2^-18 |lse|.
        margin = 1.0f - (1.52587890625e-05f + 3.814697265625e-06f * fabsf(lse));
      }
    }
    const uint64_t obase = (uint64_t)b * (uint64_t)((uint32_t)a.M * (uint32_t)a.L) + l;

    // ---- the draws, on ALL 64 lanes. Only the masked positions of the tile draw anything (an unmasked one copies its
    // token, :1203), and over a decode half of the positions are unmasked: with lane = position those lanes idle through
    // the ~550 issue cycles of a draw (one Philox block + 5 log + 5 rcp). So the masked positions park their parameters
    // in LDS by rank, the (masked position, candidate) pairs are dealt densely to the lanes, and the drawn tokens come
    // back through an LDS byte table to the lane that owns the position, which stores them coalesced as before. Which
    // lane draws a token does not matter: the Philox counter is keyed by (global position, step, m).
    const unsigned long long bal = __ballot(masked);
    const int k = __popcll(bal);                                         // masked positions in this tile
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
    float* pw = prm[wv];
    if (masked) {
#pragma unroll
      for (int v = 0; v < V; ++v) { pw[v * WAVE + rank] = qf[v]; pw[(7 + v) * WAVE + rank] = z[v]; }
      pw[5 * WAVE + rank] = fast_ok ? margin : -1.0f;                    // < 0: this position always takes the exact path
      pw[6 * WAVE + rank] = __int_as_float(lane);
      if (REPLAY) {
        pw[12 * WAVE + rank] = __int_as_float((int)b + (int)a.row_offset); pw[13 * WAVE + rank] = __int_as_float((int)l);
      } else {
        const uint64_t pos = (a.row_offset + (uint64_t)b) * (uint64_t)a.L + (uint64_t)l;
        pw[12 * WAVE + rank] = __int_as_float((int)(uint32_t)pos); pw[13 * WAVE + rank] = __int_as_float((int)(uint32_t)(pos >> 32));
      }
    }
    const int Mloc = (a.M - s0 + a.msplit - 1) / a.msplit;               // candidates of this unit: m = s0 + i * msplit
    const float inv_k = 1.0f / (float)max(k, 1);
    // coalesced write-out of candidate mi of the chunk: lane = position again
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    uint64_t ow = obase + (uint64_t)s0 * (uint64_t)a.L;                  // output offset of the next candidate to write
    const uint64_t ow_step = (uint64_t)a.msplit * (uint64_t)a.L;
    auto write_out = [&](int mi) {                                       // candidates are written in order: ow runs along
      const int tk = (int)tokw[mi * WAVE + lane];                        // (stale but in range for an unmasked lane)
      const int c = masked ? tk : xt;
      a.cand[ow] = (uint8_t)c;
      const float4 t = oh_lut[c];                                        // transform_samples, :1462-1470
      // streaming store: the one-hot is 94 % of the bytes K1 moves and is read once, by another kernel
      __builtin_nontemporal_store(f32x4_t{t.x, t.y, t.z, t.w}, reinterpret_cast<f32x4_t*>(a.onehot) + ow);
      ow += ow_step;
    };
    for (int mc = 0; mc < Mloc; mc += CR) {                              // chunks of <= 64 candidates (token table size)
      const int Mc = min(CR, Mloc - mc);
      const int T = k * Mc;
      // The pairs are dealt candidate-major (wi = mi * k + rank), and a candidate is written out as soon as its last pair
      // is drawn: its stores are in flight while the wave draws the next ones. With all draws first and all stores after,
      // every wave of the chip alternated between a VALU-only and a store-only phase in step with the others, and the
      // launch took draw time PLUS store time (137 us at B = 16384, 50 % masked, against a 76 us fill of the same bytes).
      int written = 0;
      for (int base = 0; base < T; base += WAVE) {
        const int wi = base + lane;
        if (wi < T) {
          int mi = (int)(((float)wi + 0.5f) * inv_k);                   // wi / k (T <= 4096: exact in fp32 up to the fix-up)
          int p = wi - mi * k;
          const int adj = p < 0 ? -1 : (p >= k ? 1 : 0);                // (selects, not branches: a divergent branch costs
          mi += adj; p -= adj * k;                                       //  more than the few VALU ops it skips)
          const int m = s0 + (mc + mi) * a.msplit;
          const int src = __float_as_int(pw[6 * WAVE + p]);              // the lane that owns this position
          const float mg = pw[5 * WAVE + p];
          float qv[V];
#pragma unroll
          for (int v = 0; v < V; ++v) qv[v] = pw[v * WAVE + p];
          const uint32_t key0 = (uint32_t)__float_as_int(pw[12 * WAVE + p]), key1 = (uint32_t)__float_as_int(pw[13 * WAVE + p]);
          float u[V];
          if (REPLAY) {
            const float* ub = a.uniforms + (uint64_t)m * ((uint64_t)a.u_rows * (uint64_t)a.L) * V;
#pragma unroll
            for (int v = 0; v < V; ++v) u[v] = ub[at(a.ulayout, key0, key1, v, a.L)];
          } else {
            philox_uniform5(a.seed, (uint64_t)key0 | ((uint64_t)key1 << 32), a.step, (uint32_t)m, u);
          }
          int c = 0;
          bool decided = false;
          if (mg >= 0.0f) {
            float best = -1.0f, second = -1.0f;
            int bi = 0;
#pragma unroll
            for (int v = 0; v < V; ++v) {                                // first maximum wins (:33), runner-up for the margin
              const float g = 1e-10f - log_fast(u[v] + 1e-10f);
              const float r = qv[v] * __builtin_amdgcn_rcpf(g);
              const bool gt = r > best;
              second = gt ? best : fmaxf(second, r);
              bi = gt ? v : bi;
              best = gt ? r : best;
            }
            decided = (best > 1e-30f) && (second < best * mg);
            c = bi;
          }
          ++n_draws;
          if (!decided) {                        // rare (~1e-6 of draws): exact arithmetic for this draw
            ++n_exact;
            float zs[V];
#pragma unroll
            for (int v = 0; v < V; ++v) zs[v] = pw[(7 + v) * WAVE + p];
            if (QGIVEN) {
              c = sample_categorical_1(zs, u);
            } else {
              float q[V];
              q_exact(zs, MASK, a.dm, a.mcs, q);
              c = sample_categorical_1(q, u);
            }
          }
          tokw[mi * WAVE + src] = (uint8_t)c;
        }
        const int done = min(Mc, (base + WAVE) / k);                     // candidates whose pairs are all drawn (k > 0 here)
        if (valid)
          for (int mi = written; mi < done; ++mi) write_out(mi);
        written = done;
      }
      if (valid)
        for (int mi = written; mi < Mc; ++mi) write_out(mi);             // tiles without a masked position: copies
    }
  }
  if (a.stats) {                               // soak / profiling only (wave-uniform branch)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { n_draws += __shfl_xor(n_draws, off, WAVE); n_exact += __shfl_xor(n_exact, off, WAVE); }
    if (lane == 0) { atomicAdd(&a.stats[0], (unsigned long long)n_draws); atomicAdd(&a.stats[1], (unsigned long long)n_exact); }
  }
}

// Self-test of the fast-math error bounds K1's filter relies on. Thread t sweeps a strided subset:
//  out[0][t] max relative error of g~ = 1e-10 - log_fast(u + 1e-10) vs the exact g over ALL 2^24 uniforms
//  out[1][t] max relative error of exp_fast(x) vs correctly-rounded exp over 2^24 points of [-80, 0]
//  out[2][t] max relative error of log_fast(s) vs correctly-rounded log over 2^24 points of [1, 4]
__global__ void selftest_kernel(double* out, int nthreads) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  double e0 = 0.0, e1 = 0.0, e2 = 0.0;
  for (uint32_t k = (uint32_t)t; k < (1u << 24); k += (uint32_t)nthreads) {
    const float u = (float)k * (1.0f / 16777216.0f);
    const float a = u + 1e-10f;
    const float gx = 1e-10f - logf_cr(a);
    const float gf = 1e-10f - log_fast(a);
    e0 = fmax(e0, fabs((double)gf - (double)gx) / (double)gx);
    const float x = -80.0f * u;
    const double ex = exp((double)x);
    e1 = fmax(e1, fabs((double)exp_fast(x) - ex) / ex);
    const float s = 1.0f + 3.0f * u;
    const double lx = log((double)s);
    if (k) e2 = fmax(e2, fabs((double)log_fast(s) - lx) / lx);
  }
  out[t] = e0; out[nthreads + t] = e1; out[2 * nthreads + t] = e2;
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
  // exact work-skipping (svdd_select_compact): scores holds only the LIVE candidates, slot[b*M + m] is a candidate's
  // position in it or -1 for a candidate that is a copy of its parent (its score is the parent's)
  const int32_t* slot; const float* parent_score; float* sel_score; int32_t* changed;
  int ld;   // bytes between two candidate rows of `cand` (L, or more: SVDD_OPT_CAND_ROW_STRIDE — rows padded to whole cache lines)
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
    float sv = -INFINITY;
    if (c < nchunk && m < a.M) {
      if (a.slot) { const int sl = a.slot[(int64_t)row * a.M + m]; sv = sl >= 0 ? a.scores[sl] : a.parent_score[row]; }
      else sv = s[m];
    }
    e[c] = sv;
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
  if (a.slot && lane == 0) {
    const int sl = a.slot[(int64_t)row * a.M + best];
    if (a.sel_score) a.sel_score[row] = sl >= 0 ? a.scores[sl] : a.parent_score[row];   // the next step's parent score
    if (a.changed) a.changed[row] = sl >= 0 ? 1 : 0;                                      // x_next != x
  }

  // gather the winning candidate row (index-gather compaction, diffusion_gosai.py:1226-1227)
  const uint8_t* src = a.cand + ((int64_t)row * a.M + best) * a.ld;
  uint8_t* dst = a.x_next + (int64_t)row * a.L;
  if ((a.L & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst);
    for (int i = lane; i < (a.L >> 2); i += WAVE) d4[i] = s4[i];
  } else {
    for (int i = lane; i < a.L; i += WAVE) dst[i] = src[i];
  }
}

// K2 for M <= 64 (every BASELINE config: M = 2, 10, 20): SEVERAL ROWS PER WAVE. With one wave per row, 10 of 64 lanes
// did the work and every row paid a whole wave's issue slots for the fp64 exp and the ordered sum: 212 us for 2^18 rows
// (0.07 of the HBM roofline, profiles/r03_resample_before.txt). Here a wave owns G = 64 / MP consecutive rows (MP = M
// rounded up to a power of two), lane = (row g, candidate m): the max / argmax reductions stay inside the MP-lane group
// (xor shuffles with offsets < MP), the normaliser is still summed in candidate order (one bpermute per candidate, all
// groups at once), and the G winning rows are copied with 8-byte units spread over all 64 lanes.
// Argmax shortcut (exact): when a row's best score leads the runner-up by >= 2^-18, e_best = exp(0) = 1 and every other
// e <= 1 - 2^-19, so after the common factor 1/sum the best candidate's soft value is strictly the largest — the softmax
// need not be evaluated to know its argmax. Rows closer than that (the near-uniform scores of random-init value nets,
// exact ties) take the exact path; the result is the same bits either way.
// max over the MP-lane group of a lane (MP a power of two <= 64); every lane of the group gets it. Steps below 16 are
// DPP lane permutations inside a 16-lane row (no LDS crossbar, no address arithmetic): xor 1 and 2 as quad_perm, then
// row_half_mirror / row_mirror, which after the quad steps pair every quad with the one it still lacks.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int MP>
__device__ __forceinline__ float group_max(float v) {
  if constexpr (MP >= 2) v = fmaxf(v, dpp_mov<0xB1>(v));        // quad_perm [1,0,3,2]
  if constexpr (MP >= 4) v = fmaxf(v, dpp_mov<0x4E>(v));        // quad_perm [2,3,0,1]
  if constexpr (MP >= 8) v = fmaxf(v, dpp_mov<0x141>(v));       // row_half_mirror
  if constexpr (MP >= 16) v = fmaxf(v, dpp_mov<0x140>(v));      // row_mirror
  if constexpr (MP >= 32)                                      // lane ^ 16: ds_swizzle (xor mask 0x10), no bpermute address / LDS round trip
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x1F | (0x10 << 10))));
  if constexpr (MP >= 64) v = fmaxf(v, __shfl_xor(v, 32, WAVE));
  return v;
}
// lowest candidate index m of the lane's group whose predicate holds (the group's lanes are base .. base + MP - 1)
template <int MP>
__device__ __forceinline__ int group_first(bool pred, int base) {
  const unsigned long long mask = __ballot(pred) >> base;
  constexpr unsigned long long GM = MP == 64 ? ~0ull : ((1ull << (MP & 63)) - 1ull);
  return __ffsll((long long)(mask & GM)) - 1;
}

// Value of lane J of the lane's own MP-lane group, J a compile-time constant: ds_swizzle in bit-mask mode (lane' = (lane & AND) | J
// inside each half wave; no LDS memory is touched) — or v_readlane when the group is the whole wave.
template <int MP, int J>
__device__ __forceinline__ float group_lane(float v) {
  if constexpr (MP == 64) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), J));
  } else {
    constexpr int AND = 0x1F & ~(MP - 1);
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), AND | (J << 5)));
  }
}
// ((e_0 + e_1) + e_2) + ... over the MP lanes of the group, strictly in candidate order (the oracle's fp32 sequence). Lanes beyond
// the row's M candidates hold +0, and adding +0 to a non-negative sum is exact. The MP broadcasts are independent of each other
// and issue back to back; only the MP - 1 adds are a chain (round 4: the loop `for j < M: sum += __shfl(e, base + j)` paid one
// ds_bpermute round trip per candidate, which is what made the near-tied leg of the saturated K2 benchmark 30 % slower).
template <int MP, int J>
struct OrderedSum {
  static __device__ __forceinline__ float run(float e) { return OrderedSum<MP, J - 1>::run(e) + group_lane<MP, J>(e); }
};
template <int MP>
struct OrderedSum<MP, 0> {
  static __device__ __forceinline__ float run(float e) { return group_lane<MP, 0>(e); }
};

// MS = how many lanes of the group the ordered sum visits: MP in general (lanes beyond M hold +0), exactly M for the widths
// the BASELINE configs run (M = 10, 20): 6 / 12 fewer broadcast + add pairs on the exact path.
// The decision of the G * R rows row0 .. of one wave (scores -> first-index argmax | exact softmax argmax | Philox multinomial; writes
// soft / idx / sel_score / changed) -> lane q < R * G holds the winning candidate of row row0 + q.
template <int MP, int R, int MS>
__device__ __forceinline__ int select_decide(const SelectArgs& a, const int64_t row0, const int lane) {
  constexpr int G = WAVE / MP;
  const int g = lane / MP, m = lane % MP, base = lane & ~(MP - 1);
  float svr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t row = row0 + r * G + g;
    svr[r] = -INFINITY;
    if (row < a.B && m < a.M) {
      if (a.slot) { const int sl = a.slot[row * a.M + m]; svr[r] = sl >= 0 ? a.scores[sl] : a.parent_score[row]; }
      else svr[r] = a.scores[row * a.M + m];
    }
  }
  int wbest = 0;                                            // lane q < R * G: the winning candidate of row row0 + q
#pragma unroll
  for (int r = 0; r < R; ++r) {
  const int64_t row = row0 + r * G + g;
  const bool rv = row < a.B, valid = rv & (m < a.M);
  const float sv = svr[r];
  // first-index argmax of the raw scores, and the runner-up
  const float mx = group_max<MP>(sv);
  const int bi = group_first<MP>(valid & (sv == mx), base);
  const float s2 = group_max<MP>((valid & (m != bi)) ? sv : -INFINITY);
  int best = bi;
  const bool clear = !rv | (mx - s2 >= 3.814697265625e-06f);                  // 2^-18; also M == 1 (s2 = -inf)
  const bool exact = (a.mode != SVDD_SELECT_ARGMAX) | (a.soft != nullptr) | !clear;
  if (__any(exact)) {
    const float e = valid ? expf_cr_nonpos(sv - mx) : 0.0f;
    const float sum = OrderedSum<MP, MS - 1>::run(e);                         // candidate order: ((e0 + e1) + e2) + ...
    const float rr = __fdiv_rn(1.0f, sum);
    const float p = e * rr;                                                   // ATen CPU softmax: e * (1 / sum)
    if (a.mode == SVDD_SELECT_ARGMAX) {
      const float pm = group_max<MP>(valid ? p : -INFINITY);
      best = group_first<MP>(valid & (p == pm), base);
    } else {
      const uint64_t grow = a.row_offset + (uint64_t)(rv ? row : 0);
      uint32_t ctr[4] = {(uint32_t)grow, (uint32_t)(grow >> 32), (a.step << 16), 2u};
      philox4x32_10(ctr, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
      const float u = u24(ctr[0]);
      float tot = 0.0f;
      for (int j = 0; j < a.M; ++j) { const float pj = __shfl(p, base + j, WAVE); tot = j == 0 ? pj : tot + pj; }
      const float thr = u * tot;
      float run = 0.0f;
      best = a.M - 1;
      bool found = false;
      for (int j = 0; j < a.M; ++j) {
        const float pj = __shfl(p, base + j, WAVE);
        run = j == 0 ? pj : run + pj;
        const bool hit = !found & (thr < run);
        best = hit ? j : best;
        found |= hit;
      }
    }
    if (a.soft && valid) a.soft[row * a.M + m] = p;
  }
  best = best < 0 ? 0 : best;                               // (all-NaN scores: no maximum; stay inside the row's candidates)
  if (rv && m == 0) {
    if (a.idx) a.idx[row] = best;
    if (a.slot) {
      const int sl = a.slot[row * a.M + best];
      if (a.sel_score) a.sel_score[row] = sl >= 0 ? a.scores[sl] : a.parent_score[row];   // the next step's parent score
      if (a.changed) a.changed[row] = sl >= 0 ? 1 : 0;                                      // x_next != x
    }
  }
  const int q = lane - r * G;                               // lane r G + q' takes the winner of group q' of round r
  const int bq = __shfl(best, (q >= 0 && q < G ? q : 0) * MP, WAVE);
  wbest = (q >= 0 && q < G) ? bq : wbest;
  }
  return wbest;
}

// index-gather compaction (diffusion_gosai.py:1226-1227) of the R * G winning rows of a wave in units of 8 bytes: four gathers per
// lane are REQUESTED (select_gather_issue) and stored later (select_gather_store), so that the caller can put other work — the next
// batch's decision — under their latency.
struct SelectGather { uint2 v[4]; uint8_t* dst[4]; bool ok[4]; };

template <int RG>
__device__ __forceinline__ void select_gather_issue(const SelectArgs& a, const int64_t row0, const int wbest, const int lane, const int i0,
                                                    const int U, const int total, const float inv_u, SelectGather& s) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = i0 + k * WAVE + lane;
    int gq = (int)(((float)i + 0.5f) * inv_u);            // i / U (i < 2^22: exact after the half-unit offset)
    gq = gq < RG ? gq : RG - 1;
    const int bq = __shfl(wbest, gq, WAVE);
    const int64_t rq = row0 + gq;
    s.ok[k] = i < total && rq < a.B;
    const int c = i - gq * U;
    s.dst[k] = a.x_next + rq * a.L + (int64_t)c * 8;
    s.v[k] = uint2{0u, 0u};
    if (s.ok[k]) s.v[k] = *reinterpret_cast<const uint2*>(a.cand + (rq * a.M + bq) * a.ld + (int64_t)c * 8);
  }
}
__device__ __forceinline__ void select_gather_store(const SelectGather& s) {
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (s.ok[k]) *reinterpret_cast<uint2*>(s.dst[k]) = s.v[k];
}

// NB = batches of G * R rows per wave (round 6 experiment, SVDD_OPT_SELECT_BATCHES): with one batch every wave of a saturated launch
// decides, then gathers — the decision's arithmetic (9 - 15 us of the 26 - 30 at 2^18 rows, profiles/r06_k2_gather_split.txt) and the
// gather's traffic (the rest: at the memory system's speed) add up. A wave with NB > 1 batches requests the row gathers of batch b and
// decides batch b + 1 while they are in flight: same decisions, same bytes — and SLOWER (35.8 / 41.5 us at NB = 2 / 4 against 27.4):
// fewer waves, fewer loads in flight. NB = 1 is what runs.
template <int MP, int R, int MS = MP, int NB = 1>
__global__ __launch_bounds__(256) void select_rows_kernel(SelectArgs a) {
  // R row GROUPS per wave (round 4): with one group a wave had one 512-byte gather in flight at a time, 8192 waves on the chip
  // = 4 MB in flight against the ~12 MB that 5.8 TB/s x 2 us of latency need. A wave loads the scores of its R groups back to
  // back, decides them, and issues the row gathers of all R * G winners in batches of four loads before the first store.
  constexpr int G = WAVE / MP;
  static_assert(R * G <= WAVE, "one lane per winning row");
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t row_first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * (G * R) * NB;
  if (row_first >= a.B) return;
  const uintptr_t al = (uintptr_t)(a.L | a.ld) | reinterpret_cast<uintptr_t>(a.cand) | reinterpret_cast<uintptr_t>(a.x_next);
  const int ub = (al & 7) == 0 ? 8 : (al & 3) == 0 ? 4 : (al & 1) == 0 ? 2 : 1;
  const int U = a.L / ub, total = R * G * U;
  const float inv_u = 1.0f / (float)U;
  int wbest = select_decide<MP, R, MS>(a, row_first, lane);
#pragma unroll 1
  for (int b = 0; b < NB; ++b) {
    const int64_t row0 = row_first + (int64_t)b * (G * R);
    if (row0 >= a.B) return;
    const bool more = b + 1 < NB && row0 + G * R < a.B;
    if (!a.x_next) {                                        // decision only (idx / sel_score / changed): round 6's gather-free experiment
      if (more) wbest = select_decide<MP, R, MS>(a, row0 + G * R, lane);
      continue;
    }
    if (ub == 8) {
      SelectGather sg;
      select_gather_issue<R * G>(a, row0, wbest, lane, 0, U, total, inv_u, sg);
      int wnext = 0;
      if (more) wnext = select_decide<MP, R, MS>(a, row0 + G * R, lane);     // under the gathers' latency
      select_gather_store(sg);
      for (int i0 = 4 * WAVE; i0 < total; i0 += 4 * WAVE) {
        select_gather_issue<R * G>(a, row0, wbest, lane, i0, U, total, inv_u, sg);
        select_gather_store(sg);
      }
      wbest = wnext;
      continue;
    }
    for (int i0 = 0; i0 < total; i0 += WAVE) {
      const int i = i0 + lane;
      int gq = (int)(((float)i + 0.5f) * inv_u);
      gq = gq < R * G ? gq : R * G - 1;
      const int bq = __shfl(wbest, gq, WAVE);
      const int64_t rq = row0 + gq;
      if (i < total && rq < a.B) {
        const int c = i - gq * U;
        const uint8_t* src = a.cand + (rq * a.M + bq) * a.ld + (int64_t)c * ub;
        uint8_t* dst = a.x_next + rq * a.L + (int64_t)c * ub;
        if (ub == 4) *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
        else if (ub == 2) *reinterpret_cast<uint16_t*>(dst) = *reinterpret_cast<const uint16_t*>(src);
        else *dst = *src;
      }
    }
    if (more) wbest = select_decide<MP, R, MS>(a, row0 + G * R, lane);
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
  if (a.out_f) {
#pragma unroll
    for (int v = 0; v < 4; ++v) a.out_f[(r * 4 + v) * a.L + l] = (c == v) ? 1.0f : 0.0f;
  }
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

// ------------------------------------------------- DPS: the per-position pieces around the two net passes (round 6) ----
// Gradient guidance (reference diffusion_gosai.py:1286-1330) differentiates mean(reward(softmax(E))) with respect to the one-hot
// input, E = keep * onehot(x_t) + (1 - keep) * log p(x0 | x_t) (:1325). Between the backbone and the reward net that is per-position
// arithmetic on 5 values; rounds 4-5 left it to ~40 torch element-wise kernels and their autograd twins per step. Three kernels:
//   dps_probs_kernel      logits, x -> softmax(E)[..., 0:4]                            (the reward net's input, :1326-1328)
//   dps_probs_bwd_kernel  d loss / d probs4 -> d loss / d logits (masked rows; the gradient kernel's input) and the DIRECT term
//                         keep * dE (the path through `keep * x_onehot`, unmasked rows)
//   dps_guided_q_kernel   q_xs * exp(scale * (grad - grad[MASK]))                      (:1306-1314; grad = backbone term + direct term)
// logits: the one-launch backbone's raw output, [B][L][5] contiguous. log p is subs_logp_1 (K7's arithmetic) in all three.
struct DpsArgs {
  const float* logits; const uint8_t* x; const float* g_in; const float* g_in2;
  float* out; float* out2; int B, L; float dm, mcs, scale;
};

__device__ __forceinline__ void dps_expected_probs(const float (&z)[V], int xt, float (&lp)[V], float (&pr)[V]) {
  subs_logp_1(z, xt, lp);
  float E[V];
#pragma unroll
  for (int v = 0; v < V; ++v) E[v] = (xt != MASK) ? (v == xt ? 1.0f : 0.0f) : lp[v];   // keep * onehot + (1 - keep) * logp
  float mx = E[0];
#pragma unroll
  for (int v = 1; v < V; ++v) mx = E[v] > mx ? E[v] : mx;
  float e[V], sm = 0.0f;
#pragma unroll
  for (int v = 0; v < V; ++v) { e[v] = expf(E[v] - mx); sm += e[v]; }
#pragma unroll
  for (int v = 0; v < V; ++v) pr[v] = e[v] / sm;
}

__global__ __launch_bounds__(256) void dps_probs_kernel(DpsArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int64_t)a.B * a.L) return;
  float z[V], lp[V], pr[V];
#pragma unroll
  for (int v = 0; v < V; ++v) z[v] = a.logits[n * V + v];
  dps_expected_probs(z, a.x[n], lp, pr);
  reinterpret_cast<float4*>(a.out)[n] = make_float4(pr[0], pr[1], pr[2], pr[3]);
}

__global__ __launch_bounds__(256) void dps_probs_bwd_kernel(DpsArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int64_t)a.B * a.L) return;
  const int xt = a.x[n];
  float z[V], lp[V], pr[V];
#pragma unroll
  for (int v = 0; v < V; ++v) z[v] = a.logits[n * V + v];
  dps_expected_probs(z, xt, lp, pr);
  const float4 g4 = reinterpret_cast<const float4*>(a.g_in)[n];
  const float dp[V] = {g4.x, g4.y, g4.z, g4.w, 0.0f};            // probs[..., 4] is not an input of the reward net
  float s = 0.0f;
#pragma unroll
  for (int v = 0; v < V; ++v) s += pr[v] * dp[v];
  float dE[V];
#pragma unroll
  for (int v = 0; v < V; ++v) dE[v] = pr[v] * (dp[v] - s);       // softmax backward
  if (xt != MASK) {                                               // keep = 1: E does not depend on the logits
#pragma unroll
    for (int v = 0; v < V; ++v) { a.out[n * V + v] = 0.0f; a.out2[n * V + v] = dE[v]; }
  } else {                                                        // keep = 0: log p = z' - logsumexp(z')
    float gs = 0.0f;
#pragma unroll
    for (int v = 0; v < V; ++v) gs += dE[v];
#pragma unroll
    for (int v = 0; v < V; ++v) { a.out[n * V + v] = dE[v] - expf(lp[v]) * gs; a.out2[n * V + v] = 0.0f; }
  }
}

__global__ __launch_bounds__(256) void dps_guided_q_kernel(DpsArgs a) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= (int64_t)a.B * a.L) return;
  float z[V], lp[V], g[V];
#pragma unroll
  for (int v = 0; v < V; ++v) { z[v] = a.logits[n * V + v]; g[v] = a.g_in[n * V + v] + a.g_in2[n * V + v]; }
  subs_logp_1(z, a.x[n], lp);
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const float q = v == MASK ? a.mcs : expf(lp[v]) * a.dm;      // :1306-1307, :1312 (torch.exp, not the sampler's correctly rounded one)
    a.out[n * V + v] = q * expf(a.scale * (g[v] - g[MASK]));      // :1311, :1314
  }
}

// -------------------------------------------------------------------- K4 TDS resample ----
// numpy's pairwise float32 sum (np.add.reduce), the order `ratio.sum()` uses at :1282: the array is halved (left half
// rounded down to a multiple of 8) until a block has <= 128 elements; a block is summed with 8 running accumulators.
// The blocks are independent, so they can be summed in parallel and combined in the recursion's order — same bits.
__device__ __forceinline__ float np_leaf_sum_f32(const float* p, int n) {
  if (n < 8) {
    float res = 0.0f;
    for (int i = 0; i < n; ++i) res += p[i];
    return res;
  }
  float r[8];
  for (int k = 0; k < 8; ++k) r[k] = p[k];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int k = 0; k < 8; ++k) r[k] += p[i + k];
  float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += p[i];
  return res;
}

// Walks the recursion over [0, n) in evaluation order (explicit stack: sizes halve, depth <= 40); leaf(off, len) is
// called once per block, left to right, and returns the block's sum.
template <class LeafFn>
__device__ float np_pairwise_walk(int64_t n, LeafFn leaf) {
  struct Frame { int64_t off, n; };            // n < 0: combine marker (adds the two top values, left below right)
  Frame stack[48];
  float vals[48];
  int sp = 0, vp = 0;
  stack[sp++] = {0, n};
  while (sp > 0) {
    const Frame f = stack[--sp];
    if (f.n < 0) {
      const float right = vals[--vp];
      const float left = vals[--vp];
      vals[vp++] = left + right;
    } else if (f.n <= 128) {
      vals[vp++] = leaf(f.off, (int)f.n);
    } else {
      int64_t n2 = f.n / 2;
      n2 -= n2 % 8;
      stack[sp++] = {0, -1};
      stack[sp++] = {f.off + n2, f.n - n2};      // right (evaluated second)
      stack[sp++] = {f.off, n2};                 // left (evaluated first)
    }
  }
  return vals[0];
}

struct TdsArgs {
  const float* num; const float* den; double alpha; const uint8_t* sample; const double* u;
  int B, L; uint8_t* x_next; int32_t* idx; double* work;
};

constexpr int TDS_LEAF_CAP = 2048;               // blocks of 64..128 elements: B <= 131072 in parallel, beyond that serially

// Phase 1 — ONE workgroup: the normalised CDF of np.random.choice, in numpy's summation orders (:1280-1282).
//   ratio (parallel, correctly rounded exp) -> ratio.sum() (pairwise: blocks in parallel, combined in recursion order by
//   one lane) -> p = ratio / tot widened to f64 (parallel) -> cumsum: an inherently serial chain of B float64 adds, run by
//   one wave on values it already holds in registers (64 coalesced loads, then 64 readlane + add steps; lane j keeps the
//   j-th prefix; the next 64 values are in flight meanwhile) -> cdf /= cdf[-1] (parallel).
// Round 2 ran the two serial loops on one lane straight from global memory: ~0.3 us per element (0.6 ms at B = 2048,
// 28.8 ms at B = 65536, profiles/r03_resample_before.txt).
__global__ __launch_bounds__(1024) void tds_cdf_kernel(TdsArgs a) {
  __shared__ int leaf_off[TDS_LEAF_CAP];
  __shared__ int leaf_len[TDS_LEAF_CAP];
  __shared__ float leaf_val[TDS_LEAF_CAP];
  __shared__ int nleaf_s;
  __shared__ float tot_s;
  float* ratio = reinterpret_cast<float*>(a.work + a.B);   // work: [B] f64 cdf + [B] f32 ratio
  const float inv_alpha = (float)(1.0 / a.alpha);           // the Python double 1.0/alpha, rounded to fp32 once
  for (int b = threadIdx.x; b < a.B; b += blockDim.x)
    ratio[b] = expf_cr(inv_alpha * (a.num[b] - a.den[b]));  // :1280
  if (threadIdx.x == 0) {
    int n = 0;
    np_pairwise_walk(a.B, [&](int64_t off, int len) {
      if (n < TDS_LEAF_CAP) { leaf_off[n] = (int)off; leaf_len[n] = len; }
      ++n;
      return 0.0f;
    });
    nleaf_s = n;
  }
  __syncthreads();
  const int nleaf = nleaf_s;
  if (nleaf <= TDS_LEAF_CAP) {
    for (int k = threadIdx.x; k < nleaf; k += blockDim.x) leaf_val[k] = np_leaf_sum_f32(ratio + leaf_off[k], leaf_len[k]);
    __syncthreads();
    if (threadIdx.x == 0) {
      int k = 0;
      tot_s = np_pairwise_walk(a.B, [&](int64_t, int) { return leaf_val[k++]; });
    }
  } else if (threadIdx.x == 0) {
    tot_s = np_pairwise_walk(a.B, [&](int64_t off, int len) { return np_leaf_sum_f32(ratio + off, len); });
  }
  __syncthreads();
  const float tot = tot_s;                                                       // ratio.sum()
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) a.work[b] = (double)__fdiv_rn(ratio[b], tot);   // p (f32) -> f64
  __syncthreads();
  if (threadIdx.x < WAVE) {                                                      // p.cumsum(): c_j = c_{j-1} + p_j
    const int lane = threadIdx.x;
    double c = 0.0;
    double pcur = lane < a.B ? a.work[lane] : 0.0;
    for (int base = 0; base < a.B; base += WAVE) {
      const int nx = base + WAVE + lane;
      const double pnext = nx < a.B ? a.work[nx] : 0.0;
      const int lo = __double2loint(pcur), hi = __double2hiint(pcur);
      double keep = 0.0;
#pragma unroll
      for (int j = 0; j < WAVE; ++j) {
        c += __hiloint2double(__builtin_amdgcn_readlane(hi, j), __builtin_amdgcn_readlane(lo, j));   // (+ 0.0 past the end)
        if (lane == j) keep = c;
      }
      if (base + lane < a.B) a.work[base + lane] = keep;
      pcur = pnext;
    }
  }
  __syncthreads();
  const double last = a.work[a.B - 1];
  __syncthreads();
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) a.work[b] = a.work[b] / last;   // cdf /= cdf[-1]
}

// Phase 2 — the whole chip: lane = particle: searchsorted(cdf, u, side='right') (:1282), then the wave copies its 64
// ancestors' rows x_next[j] = sample[idx[j]] (:1284) in 8-byte units spread over all lanes.
__global__ __launch_bounds__(256) void tds_gather_kernel(TdsArgs a) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t j0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * WAVE;
  if (j0 >= a.B) return;
  const int64_t j = j0 + lane;
  int k = 0;
  if (j < a.B) {
    const double uj = a.u[j];
    int lo = 0, hi = a.B;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.work[mid] <= uj) lo = mid + 1; else hi = mid; }
    k = lo < a.B ? lo : a.B - 1;
    if (a.idx) a.idx[j] = k;
  }
  const uintptr_t both = reinterpret_cast<uintptr_t>(a.sample) | reinterpret_cast<uintptr_t>(a.x_next);
  const int ub = ((a.L | both) & 7) == 0 ? 8 : ((a.L | both) & 3) == 0 ? 4 : ((a.L | both) & 1) == 0 ? 2 : 1;
  const int U = a.L / ub, total = WAVE * U;
  for (int i0 = 0; i0 < total; i0 += WAVE) {
    const int i = i0 + lane;
    const int rq = i / U;                                   // < 64
    const int kq = __shfl(k, rq, WAVE);
    if (j0 + rq < a.B) {
      const int c = i - rq * U;
      const uint8_t* src = a.sample + (int64_t)kq * a.L + (int64_t)c * ub;
      uint8_t* dst = a.x_next + (j0 + rq) * a.L + (int64_t)c * ub;
      if (ub == 8) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(src);
      else if (ub == 4) *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(src);
      else if (ub == 2) *reinterpret_cast<uint16_t*>(dst) = *reinterpret_cast<const uint16_t*>(src);
      else *dst = *src;
    }
  }
}

// ---------------------------------------------------------------- exact work-skipping: compaction helpers ----
// A candidate that unmasked nothing IS its parent (diffusion_gosai.py:1203 copies x_t where nothing is drawn), and with
// time_conditioning off (:334-335) every net output for it is the parent's. The engine therefore evaluates the nets only
// on the LIVE candidates / rows. The compaction runs on the device (flags -> stable prefix scan -> index list + count in
// device memory) and every consumer kernel reads the count from there: no host round trip inside the diffusion loop.
//   flags[i] != 0 -> live_idx[k] = i, slot[i] = k (k = number of live items before i) ; else slot[i] = -1 ; count[0] = #live
__global__ __launch_bounds__(1024) void compact_flags_kernel(const int32_t* __restrict__ flags, int n, int32_t* __restrict__ live_idx,
                                                             int32_t* __restrict__ slot, int32_t* __restrict__ count) {
  __shared__ int wave_tot[16];
  __shared__ int base_s;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const bool live = i < n && flags[i] != 0;
    const unsigned long long m = __ballot(live);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[w] = __popcll(m);
    __syncthreads();
    int off = base_s;
    for (int k = 0; k < w; ++k) off += wave_tot[k];
    if (i < n) {
      if (live) { live_idx[off + before] = i; slot[i] = off + before; }
      else slot[i] = -1;
    }
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wave_tot[k]; base_s += t; }
    __syncthreads();
  }
  if (threadIdx.x == 0) count[0] = base_s;
}

// Compaction ORDERED BY KEY, largest first, stable inside a key (counting sort; one workgroup): key[i] > 0 (clamped to 15) marks a
// live item. The windowed conv tower's workgroups take as many row tiles as their candidate's window has (key = that number); two
// share a CU and the dispatcher hands them out in grid order — with the long windows first the launch does not end on a few CUs
// that started a 13-tile window last (profiles/r05_tower_order_probe.txt: 487 -> 395 us per launch on the states of a C2 decode).
__global__ __launch_bounds__(1024) void compact_by_key_kernel(const int32_t* __restrict__ key, int n, int32_t* __restrict__ live_idx,
                                                              int32_t* __restrict__ slot, int32_t* __restrict__ count, int split) {
  __shared__ int hist[16], base[16];
  __shared__ int wave_cnt[16][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x < 16) hist[threadIdx.x] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const int k = i < n ? min(max(key[i], 0), 15) : 0;
#pragma unroll
    for (int kk = 1; kk < 16; ++kk) {
      const unsigned long long m = __ballot(k == kk);
      if (lane == 0 && m) atomicAdd(&hist[kk], __popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int kk = 15; kk >= 1; --kk) { base[kk] = run; run += hist[kk]; }
    count[0] = run;
    if (split > 0) { count[1] = min(run, split); count[2] = max(run - split, 0); }   // the list as two parts: [0, split) and the rest
  }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + threadIdx.x;
    const int k = i < n ? min(max(key[i], 0), 15) : 0;
    int rank = 0;
#pragma unroll
    for (int kk = 1; kk < 16; ++kk) {
      const unsigned long long m = __ballot(k == kk);
      if (k == kk) rank = __popcll(m & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[w][kk] = __popcll(m);
    }
    __syncthreads();
    if (i < n) {
      if (k > 0) {
        int off = base[k] + rank;
        for (int ww = 0; ww < w; ++ww) off += wave_cnt[ww][k];
        live_idx[off] = i; slot[i] = off;
      } else slot[i] = -1;
    }
    __syncthreads();
    if (threadIdx.x >= 1 && threadIdx.x < 16) {
      int t = 0;
      for (int ww = 0; ww < 16; ++ww) t += wave_cnt[ww][threadIdx.x];
      base[threadIdx.x] += t;
    }
    __syncthreads();
  }
}

// dst[i, :] = src[idx[i], :] for i < count : rows of `row_bytes` bytes, one wave per row
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ idx,
                                                          const int32_t* __restrict__ count, int n, int row_bytes,
                                                          uint8_t* __restrict__ dst) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= n || (count && r >= *count)) return;
  const int lane = threadIdx.x & 63;
  const uint8_t* s = src + (size_t)idx[r] * row_bytes;
  uint8_t* d = dst + (size_t)r * row_bytes;
  if ((row_bytes & 3) == 0) {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
    for (int i = lane; i < (row_bytes >> 2); i += WAVE) d4[i] = s4[i];
  } else {
    for (int i = lane; i < row_bytes; i += WAVE) d[i] = s[i];
  }
}

// The selected candidate becomes the next parent: dst[b, :] = src[slot[b*M + sel[b]], :] when that candidate was live
// (slot >= 0), else dst[b] is left as it is (the parent did not change). Rows of `row_bytes` bytes (multiple of 16).
__global__ __launch_bounds__(256) void advance_rows_kernel(const uint8_t* __restrict__ src, const int32_t* __restrict__ slot,
                                                           const int32_t* __restrict__ sel, int B, int M, int row_bytes,
                                                           uint8_t* __restrict__ dst) {
  const int b = blockIdx.x;
  const int sl = slot[(size_t)b * M + sel[b]];
  if (sl < 0) return;
  if ((row_bytes & 15) == 0) {
    const uint4* s16 = reinterpret_cast<const uint4*>(src + (size_t)sl * row_bytes);
    uint4* d16 = reinterpret_cast<uint4*>(dst + (size_t)b * row_bytes);
    for (int i = threadIdx.x; i < (row_bytes >> 4); i += 256) d16[i] = s16[i];
  } else {
    const uint32_t* s4 = reinterpret_cast<const uint32_t*>(src + (size_t)sl * row_bytes);
    uint32_t* d4 = reinterpret_cast<uint32_t*>(dst + (size_t)b * row_bytes);
    for (int i = threadIdx.x; i < (row_bytes >> 2); i += 256) d4[i] = s4[i];
  }
}


// ======================================================================== K8 mt19937 ====
// torch's CPU generator stream ON THE DEVICE (parity-mode RNG at speed). The reference draws its categorical uniforms with
// rand_like(q_xs) (diffusion_gosai.py:33) from torch's global CPU generator: at::mt19937 (= std::mt19937) + ATen's
// uniform_real_distribution<float>, u = (y & 0xFFFFFF) * 2^-24, one 32-bit output per float, sequentially. Round 3 replayed it
// on the host (torch.rand + a 10 MB upload per diffusion step). Here the 624-word state crosses once per decode and the stream
// is produced by ONE workgroup (the recurrence is serial):
//   as an infinite sequence x[n + 624] = x[n + 397] ^ tw(x[n], x[n + 1]), the saved state being x[0 .. 623] and the outputs
//   temper(x[pos + i]). x[n + 624] depends on nothing younger than x[n + 397], so 227 consecutive elements are independent:
//   one GENERATION = 227 lanes. Lane j's x[n + 397] is its own result of the generation before (a register); x[n], x[n + 1]
//   were written 2 to 3 generations earlier by other lanes (LDS ring), so ONE barrier per TWO generations orders everything;
//   each lane tempers, converts and stores its own element (coalesced 4-byte stores).
// state [625] u32 = 624 words + pos (next output index, 624 = "twist first", what torch.manual_seed leaves). On return the state
// is the last 624 words of the sequence and the matching pos (any window of the sequence is a valid state: the recurrence is
// shift-invariant), i.e. what the host generator would hold after drawing n floats, up to that rotation.
constexpr int MT_N = 624, MT_M = 397, MT_GEN = MT_N - MT_M /* 227 */, MT_RING = 2048;

__device__ __forceinline__ uint32_t mt_tw(uint32_t a, uint32_t b) {
  const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((b & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
__device__ __forceinline__ float mt_float(uint32_t x) { return (float)(mt_temper(x) & 0xFFFFFFu) * (1.0f / 16777216.0f); }

// 256 threads and <= 16 VGPRs ON PURPOSE: the backbone's 256 workgroups take every CU for 2.1 of a step's 3.2 ms with 2 x 248
// VGPRs per SIMD, which leaves 16 per SIMD and 16 KB of LDS — exactly enough for this kernel's four waves to run UNDER a backbone
// workgroup instead of waiting for a CU. (A second version with separate twister / output waves was no faster alone — 0.565 vs
// 0.579 ns per output: the interval is bound by its LDS round trip + barrier, not by the output path — and, needing 2 waves per
// SIMD, could no longer co-reside: the replay decode went from 1.17x to 1.29x of the Philox decode. profiles/r04_mt_microbench.txt)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(16))) void mt19937_kernel(uint32_t* __restrict__ state, float* __restrict__ out, long long n) {
  __shared__ uint32_t ring[MT_RING];
  const int j = threadIdx.x;
  const long long pos = (long long)state[MT_N];
  const long long total = pos + n;                        // one past the last element consumed
  for (int k = j; k < MT_N; k += 256) {
    const uint32_t v = state[k];
    ring[k] = v;
    if (k >= pos && k < total) out[k - pos] = mt_float(v);  // what is left of the current block
  }
  __syncthreads();
  // generations come in pairs; G = generations needed so that 624 + 227 G >= total
  long long G = total > MT_N ? (total - MT_N + MT_GEN - 1) / MT_GEN : 0;
  G += (G & 1);
  const bool lane = j < MT_GEN;
  uint32_t prev = lane ? ring[MT_M + j] : 0u;             // x[397 + j]: "generation -1" of this lane
  int r0 = j;                                             // ring slot of x[base + j], base = 227 g
  // element index of this lane's first output of the pair, relative to the first element wanted: i0 = 624 + base + j - pos
  long long sbase = (long long)MT_N - pos;                // wave-uniform (scalar registers): this lane's i0 = sbase + j
  for (long long g = 0; g < G; g += 2) {
    if (lane) {
      const int r1 = (r0 + MT_GEN) & (MT_RING - 1);
      const uint32_t a0 = ring[r0], b0 = ring[(r0 + 1) & (MT_RING - 1)];
      const uint32_t a1 = ring[r1], b1 = ring[(r1 + 1) & (MT_RING - 1)];
      const uint32_t x0 = prev ^ mt_tw(a0, b0);           // x[base + 624 + j]
      const uint32_t x1 = x0 ^ mt_tw(a1, b1);             // x[base + 227 + 624 + j]   (its x[n + 397] is x0)
      prev = x1;
      ring[(r0 + MT_N) & (MT_RING - 1)] = x0;
      ring[(r0 + MT_N + MT_GEN) & (MT_RING - 1)] = x1;
      r0 = (r0 + 2 * MT_GEN) & (MT_RING - 1);
      // the whole pair inside [0, n) (every interval but the first and the last few): no per-lane range arithmetic
      if (sbase >= 0 && sbase + 2 * MT_GEN <= n) {
        float* o = out + sbase;
        o[j] = mt_float(x0);
        o[j + MT_GEN] = mt_float(x1);
      } else {
        const long long i0 = sbase + j, i1 = i0 + MT_GEN;
        if (i0 >= 0 && i0 < n) out[i0] = mt_float(x0);
        if (i1 >= 0 && i1 < n) out[i1] = mt_float(x1);
      }
    }
    sbase += 2 * MT_GEN;
    // LDS-only barrier: __syncthreads() also waits for this interval's GLOBAL stores to be acknowledged (s_waitcnt vmcnt(0)) —
    // a memory round trip per 454 outputs, which is what all three kernel variants were actually measuring (0.58 - 0.62 ns per
    // output whatever the instruction count). Nobody reads `out` inside the kernel: the stores may stay in flight.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // new state: the window x[E .. E + 623], E = 227 G ; pos' = total - E (0 <= pos' <= 624)
  const long long E = G * MT_GEN;
  for (int k = j; k < MT_N; k += 256) state[k] = ring[(int)((E + k) & (MT_RING - 1))];
  if (j == 0) state[MT_N] = (uint32_t)(total - E);
}

// Optional per-launch timing (bench.py's roofline leg): when enabled, K1/K2 are launched with
// hipExtLaunchKernelGGL start/stop events, i.e. HIP events bound to the dispatch itself on the launch
// stream; svdd_profile_collect() sums hipEventElapsedTime over the recorded launches.
struct TimedLaunch { hipEvent_t start, stop; };
constexpr int PROFILE_KERNELS = 13;           // 0 propose (K1), 1 select (K2), 2 conv1d, 3 gru, 4 epilogue_ln, 5 conv_tower, 6 backbone_cnn, 7 value_tail, 8 tds_resample (K4, both phases), 9 mt19937 (K8), 10 backbone gradient, 11 GRU forward that saves its gates (DPS), 12 GRU BPTT (DPS)
bool g_profile = false;
TimedLaunch* g_timed[PROFILE_KERNELS] = {};
int g_timed_n[PROFILE_KERNELS] = {}, g_timed_cap[PROFILE_KERNELS] = {};

TimedLaunch* timed_slot(int k) {
  if (!g_profile) return nullptr;
  if (g_timed_n[k] == g_timed_cap[k]) {
    const int cap = g_timed_cap[k] ? g_timed_cap[k] * 2 : 1024;
    TimedLaunch* p = (TimedLaunch*)realloc(g_timed[k], sizeof(TimedLaunch) * (size_t)cap);
    if (!p) return nullptr;
    g_timed[k] = p; g_timed_cap[k] = cap;
  }
  TimedLaunch* t = &g_timed[k][g_timed_n[k]];
  if (hipEventCreate(&t->start) != hipSuccess) return nullptr;
  if (hipEventCreate(&t->stop) != hipSuccess) { (void)hipEventDestroy(t->start); return nullptr; }
  ++g_timed_n[k];
  return t;
}

int g_msplit = 0;        // svdd_set_option(SVDD_OPT_MSPLIT, k): override K1's candidate split (0 = auto)
int g_force_exact = 0;   // svdd_set_option(SVDD_OPT_FORCE_EXACT, 1): K1 takes the exact path for every draw
int g_cand_ld = 0;        // svdd_set_option(SVDD_OPT_CAND_ROW_STRIDE, bytes): row stride of `cand` in svdd_select* (0 = L)
int g_select_batches = 0;   // svdd_set_option(SVDD_OPT_SELECT_BATCHES, n): batches of row groups per wave in saturated svdd_select launches (0 / 1 = one, the default; 2, 4: the round-6 experiment)
int g_select_one_row_per_wave = 0;   // svdd_set_option(SVDD_OPT_SELECT_ONE_ROW, v): 1 = K2 as one wave per row for every M (A/B); 2 / 3 = the rows-per-wave kernel with 4 / 1 row groups per wave whatever the batch (0: by size)
unsigned long long* g_k1_stats = nullptr;   // svdd_k1_stats: device counters K1 adds to

inline int check_launch() { return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH; }
inline bool bad_layout(int layout) { return layout != SVDD_LAYOUT_BLV && layout != SVDD_LAYOUT_BVL; }

}  // namespace

extern "C" void svdd_internal_set_bb_lp_version(int v);      // svdd_lp_backbone.hip
extern "C" void svdd_internal_set_trunk_gemm_version(int v); // svdd_trunk.hip
extern "C" void svdd_internal_set_trunk_planes_f32(int v);   // svdd_trunk.hip
extern "C" void svdd_internal_set_bb_split(int v);           // svdd_nets.hip

// ================================================================================ C ABI ====
extern "C" {

int svdd_abi_version(void) { return SVDD_ABI_VERSION; }

int svdd_set_option(int key, int value) {
  if (key == SVDD_OPT_FORCE_EXACT) { g_force_exact = value ? 1 : 0; return SVDD_OK; }
  if (key == SVDD_OPT_MSPLIT && value >= 0 && value <= 64) { g_msplit = value; return SVDD_OK; }
  if (key == SVDD_OPT_SELECT_ONE_ROW && value >= 0 && value <= 3) { g_select_one_row_per_wave = value; return SVDD_OK; }
  if (key == SVDD_OPT_CAND_ROW_STRIDE && value >= 0) {
    // a layout EXPERIMENT: svdd_select* would stride `cand` by `value` bytes without any way to check that the caller's buffer is
    // padded that way (out-of-bounds reads otherwise) — honoured only in a process that opted in (tools/resample_microbench.py does)
    if (value != 0 && !getenv("SVDD_EXPERIMENTS")) return SVDD_E_ARG;
    g_cand_ld = value; return SVDD_OK;
  }
  if (key == SVDD_OPT_SELECT_BATCHES && value >= 0 && value <= 5) { g_select_batches = value; return SVDD_OK; }
  if (key == SVDD_OPT_TRUNK_PLANES_F32) { svdd_internal_set_trunk_planes_f32(value); return SVDD_OK; }
  if (key == SVDD_OPT_BACKBONE_SPLIT) { svdd_internal_set_bb_split(value); return SVDD_OK; }
  if (key == SVDD_OPT_BACKBONE_LP_VERSION) { svdd_internal_set_bb_lp_version(value); return SVDD_OK; }
  if (key == SVDD_OPT_TRUNK_GEMM_VERSION) { svdd_internal_set_trunk_gemm_version(value); return SVDD_OK; }
  return SVDD_E_ARG;
}

int svdd_k1_stats(unsigned long long* device_counters2) {
  g_k1_stats = device_counters2;              // NULL switches the counting off again
  return SVDD_OK;
}

// used by svdd_nets.hip: start/stop events for a timed launch of net kernel k (nullptrs when profiling is off)
void svdd_internal_timed_events(int k, hipEvent_t* e0, hipEvent_t* e1) {
  TimedLaunch* t = timed_slot(k);
  *e0 = t ? t->start : nullptr;
  *e1 = t ? t->stop : nullptr;
}

int svdd_profile_enable(int on) {
  g_profile = on != 0;
  return SVDD_OK;
}

int svdd_profile_collect(int kernel, double* total_ms, int* launches) {
  if (kernel < 0 || kernel >= PROFILE_KERNELS || !total_ms || !launches) return SVDD_E_ARG;
  double tot = 0.0;
  int n = 0;
  for (int i = 0; i < g_timed_n[kernel]; ++i) {
    TimedLaunch& t = g_timed[kernel][i];
    float ms = 0.0f;
    if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      tot += ms; ++n;
    }
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  g_timed_n[kernel] = 0;
  *total_ms = tot;
  *launches = n;
  return SVDD_OK;
}

int svdd_selftest_fastmath(double* out3) {
  if (!out3) return SVDD_E_ARG;
  const int nthreads = 256 * 1024;
  double* d = nullptr;
  if (hipMalloc(&d, sizeof(double) * 3 * nthreads) != hipSuccess) return SVDD_E_NODEVICE;
  hipLaunchKernelGGL(selftest_kernel, dim3(1024), dim3(256), 0, 0, d, nthreads);
  double* h = (double*)malloc(sizeof(double) * 3 * nthreads);
  const bool ok = hipMemcpy(h, d, sizeof(double) * 3 * nthreads, hipMemcpyDeviceToHost) == hipSuccess;
  if (ok)
    for (int j = 0; j < 3; ++j) {
      double m = 0.0;
      for (int i = 0; i < nthreads; ++i) m = h[j * nthreads + i] > m ? h[j * nthreads + i] : m;
      out3[j] = m;
    }
  free(h);
  (void)hipFree(d);
  return ok ? SVDD_OK : SVDD_E_LAUNCH;
}

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

static inline int replay_rows(const svdd_rng_t* rng, int B) { return rng->uniforms_rows > 0 ? rng->uniforms_rows : B; }

static int launch_propose(bool q_given, const float* logits, const uint8_t* x, float dm, float mcs, int B, int L, int M,
                          int layout, const svdd_rng_t* rng, uint8_t* cand, float* onehot, float* q_xs, void* stream);

int svdd_propose(const float* logits, const uint8_t* x, float dm, float mcs, int B, int L, int M, int layout,
                 const svdd_rng_t* rng, uint8_t* cand, float* onehot, float* q_xs, void* stream) {
  return launch_propose(false, logits, x, dm, mcs, B, L, M, layout, rng, cand, onehot, q_xs, stream);
}

int svdd_sample_categorical(const float* q, const uint8_t* x, int B, int L, int M, int layout, const svdd_rng_t* rng,
                            uint8_t* cand, float* onehot, void* stream) {
  return launch_propose(true, q, x, 0.0f, 0.0f, B, L, M, layout, rng, cand, onehot, nullptr, stream);
}

static int launch_propose(bool q_given, const float* logits, const uint8_t* x, float dm, float mcs, int B, int L, int M,
                          int layout, const svdd_rng_t* rng, uint8_t* cand, float* onehot, float* q_xs, void* stream) {
  if (!logits || !x || !rng || !cand || !onehot || B <= 0 || L <= 0 || M <= 0 || M > 65535 || bad_layout(layout))
    return SVDD_E_ARG;
  if (rng->kind == SVDD_RNG_REPLAY ? (rng->uniforms == nullptr || bad_layout(rng->uniforms_layout))
                                   : rng->kind != SVDD_RNG_PHILOX) return SVDD_E_ARG;
  const int64_t N = (int64_t)B * L;
  if (N >= (int64_t)1 << 31 || (int64_t)M * L >= (int64_t)1 << 32) return SVDD_E_ARG;
  const int64_t ntiles = (N + WAVE - 1) / WAVE;
  // split the M candidates of a tile over up to 4 waves only while the chip (256 CUs x 4 SIMDs) is underfilled
  int msplit = g_msplit > 0 ? g_msplit : (ntiles >= 4096 ? 1 : ntiles >= 256 ? 2 : 4);   // measured: tools/k1_microbench.py
  if (msplit > M) msplit = M;
  const int mloc = (M + msplit - 1) / msplit;
  const int tok_rows = mloc < WAVE ? mloc : WAVE;
  ProposeArgs a{logits, x, dm, mcs, B, L, M, layout, rng->kind, rng->step, rng->uniforms, rng->seed,
                rng->row_offset, cand, onehot, q_xs, g_force_exact, msplit, rng->uniforms_layout, g_k1_stats, tok_rows,
                (replay_rows(rng, B))};
  if (rng->kind == SVDD_RNG_REPLAY && (a.u_rows < B || rng->row_offset + (uint64_t)B > (uint64_t)a.u_rows ||
                                        (int64_t)a.u_rows * L >= (int64_t)1 << 31)) return SVDD_E_ARG;
  if (rng->kind == SVDD_RNG_REPLAY && rng->uniforms_rows <= 0) a.row_offset = 0;   // plain replay: the blocks are this batch's own
  const size_t lds = (size_t)4 * tok_rows * WAVE;
  const bool replay = rng->kind == SVDD_RNG_REPLAY;
  auto k = q_given ? (replay ? propose_kernel<true, true> : propose_kernel<false, true>)
                   : (replay ? propose_kernel<true, false> : propose_kernel<false, false>);
  // persistent blocks: exactly as many as the chip holds at once (a second, partly filled round of blocks cost 20 %)
  static int s_cus = 0;
  static int s_occ[4][WAVE + 1];                          // blocks per CU by (kernel variant, tok_rows)
  if (!s_cus) {
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SVDD_E_NODEVICE;
    s_cus = prop.multiProcessorCount;
  }
  int& occ = s_occ[(q_given ? 2 : 0) + (replay ? 1 : 0)][tok_rows];
  if (!occ) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), 256, lds) != hipSuccess || nb < 1) nb = 4;
    occ = nb;
  }
  const int64_t nblocks = (ntiles * msplit + 3) / 4;
  const int64_t cap = (int64_t)s_cus * occ;
  const unsigned grid = (unsigned)(nblocks < cap ? nblocks : cap);
  TimedLaunch* t = timed_slot(0);
  hipEvent_t e0 = t ? t->start : nullptr, e1 = t ? t->stop : nullptr;
  hipExtLaunchKernelGGL(k, dim3(grid), dim3(256), lds, (hipStream_t)stream, e0, e1, 0, a);
  return check_launch();
}

int svdd_select(const float* scores, const uint8_t* cand, int B, int L, int M, int mode, const svdd_rng_t* rng,
                uint8_t* x_next, float* soft, int32_t* idx, void* stream) {
  return svdd_select_compact(scores, nullptr, nullptr, cand, B, L, M, mode, rng, x_next, soft, idx, nullptr, nullptr, stream);
}

int svdd_select_compact(const float* scores, const int32_t* slot, const float* parent_score, const uint8_t* cand, int B,
                        int L, int M, int mode, const svdd_rng_t* rng, uint8_t* x_next, float* soft, int32_t* idx,
                        float* sel_score, int32_t* changed, void* stream) {
  if (!scores || !cand || (!x_next && !idx) || B <= 0 || L <= 0 || M <= 0 || M > SVDD_MAX_M) return SVDD_E_ARG;
  if (!x_next && !(M <= WAVE && g_select_one_row_per_wave != 1)) return SVDD_E_ARG;   // the decision-only form exists in select_rows_kernel only
  if (mode != SVDD_SELECT_ARGMAX && mode != SVDD_SELECT_MULTINOMIAL) return SVDD_E_ARG;
  if (mode == SVDD_SELECT_MULTINOMIAL && (!rng || rng->kind != SVDD_RNG_PHILOX)) return SVDD_E_ARG;
  if (slot && !parent_score) return SVDD_E_ARG;
  SelectArgs a{scores, cand, B, L, M, mode, rng ? rng->step : 0u, rng ? rng->seed : 0ull,
               rng ? rng->row_offset : 0ull, x_next, soft, idx, slot, parent_score, sel_score, changed,
               g_cand_ld >= L ? g_cand_ld : L};
  TimedLaunch* t = timed_slot(1);
  hipEvent_t e0 = t ? t->start : nullptr, e1 = t ? t->stop : nullptr;
  if (M <= WAVE && g_select_one_row_per_wave != 1) {
    int mp = 1;
    while (mp < M) mp <<= 1;
    // row groups per wave: 4 once there are enough rows to fill the chip several times over (memory-level parallelism),
    // 1 at the decode's own sizes (a few hundred rows: latency, spread over as many waves as possible)
    // (the winners of a wave's R * G rows travel in one lane each: R * G <= 64, i.e. R <= MP)
    const int Rw = g_select_one_row_per_wave == 2 ? 4 : g_select_one_row_per_wave == 3 ? 1 : ((int64_t)B * mp >= (int64_t)1 << 21 ? 4 : 1);
    const int R = Rw > mp ? mp : Rw;
    const int64_t waves = ((int64_t)B + (WAVE / mp) * R - 1) / ((WAVE / mp) * R);
    const dim3 grid((unsigned)((waves + 3) / 4));
#define SVDD_SEL_LAUNCH(MP_)                                                                                                     \
    if (R > 1) hipExtLaunchKernelGGL((select_rows_kernel<MP_, (MP_ >= 4 ? 4 : MP_)>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a); \
    else hipExtLaunchKernelGGL((select_rows_kernel<MP_, 1>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
    if (M == 10 || M == 20) {                               // the BASELINE widths: the ordered sum stops at M
      // saturated launches (R > 1), EXPERIMENT (svdd_set_option(SVDD_OPT_SELECT_BATCHES, 2 | 4)): NB batches per wave, the next batch's
      // decision under the row gathers of the one before. Measured slower (profiles/r06_k2_gather_split.txt: 27.4 -> 35.8 / 41.5 us at
      // 2^18 rows, M = 10): a quarter of the waves means a quarter of the loads in flight — the kernel lives on memory-level
      // parallelism, not on overlap inside a wave. Default: one batch per wave (round 5's launch).
      const int nb = R > 1 ? (g_select_batches == 2 ? 2 : g_select_batches == 4 ? 4 : 1) : 1;
      const dim3 gridb((unsigned)(((waves + nb - 1) / nb + 3) / 4));
      if (R > 1 && (g_select_batches == 3 || g_select_batches == 5)) {
        // the same experiment at an UNCHANGED wave count: the wave's 4 row groups as 2 batches of 2 (3) or 4 batches of 1 (5)
        if (M == 10) {
          if (g_select_batches == 3) hipExtLaunchKernelGGL((select_rows_kernel<16, 2, 10, 2>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
          else hipExtLaunchKernelGGL((select_rows_kernel<16, 1, 10, 4>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        } else {
          if (g_select_batches == 3) hipExtLaunchKernelGGL((select_rows_kernel<32, 2, 20, 2>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
          else hipExtLaunchKernelGGL((select_rows_kernel<32, 1, 20, 4>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        }
        return check_launch();
      }
      if (M == 10) {
        if (nb == 4) hipExtLaunchKernelGGL((select_rows_kernel<16, 4, 10, 4>), gridb, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        else if (nb == 2) hipExtLaunchKernelGGL((select_rows_kernel<16, 4, 10, 2>), gridb, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        else if (R > 1) hipExtLaunchKernelGGL((select_rows_kernel<16, 4, 10>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        else hipExtLaunchKernelGGL((select_rows_kernel<16, 1, 10>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
      } else {
        if (nb == 4) hipExtLaunchKernelGGL((select_rows_kernel<32, 4, 20, 4>), gridb, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        else if (nb == 2) hipExtLaunchKernelGGL((select_rows_kernel<32, 4, 20, 2>), gridb, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        else if (R > 1) hipExtLaunchKernelGGL((select_rows_kernel<32, 4, 20>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
        else hipExtLaunchKernelGGL((select_rows_kernel<32, 1, 20>), grid, dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
      }
      return check_launch();
    }
    switch (mp) {
      case 1: SVDD_SEL_LAUNCH(1) break;
      case 2: SVDD_SEL_LAUNCH(2) break;
      case 4: SVDD_SEL_LAUNCH(4) break;
      case 8: SVDD_SEL_LAUNCH(8) break;
      case 16: SVDD_SEL_LAUNCH(16) break;
      case 32: SVDD_SEL_LAUNCH(32) break;
      default: SVDD_SEL_LAUNCH(64) break;
    }
#undef SVDD_SEL_LAUNCH
    return check_launch();
  }
  hipExtLaunchKernelGGL(select_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  return check_launch();
}

int svdd_compact_flags(const int32_t* flags, int n, int32_t* live_idx, int32_t* slot, int32_t* count, void* stream) {
  if (!flags || !live_idx || !slot || !count || n <= 0) return SVDD_E_ARG;
  hipLaunchKernelGGL(compact_flags_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, flags, n, live_idx, slot, count);
  return check_launch();
}

int svdd_compact_by_key(const int32_t* key, int n, int32_t* live_idx, int32_t* slot, int32_t* count, int split, void* stream) {
  if (!key || !live_idx || !slot || !count || n <= 0 || split < 0) return SVDD_E_ARG;
  hipLaunchKernelGGL(compact_by_key_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, key, n, live_idx, slot, count, split);
  return check_launch();
}

int svdd_gather_rows(const void* src, const int32_t* idx, const int32_t* count, int n, int row_bytes, void* dst, void* stream) {
  if (!src || !idx || !dst || n <= 0 || row_bytes <= 0) return SVDD_E_ARG;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const uint8_t*)src, idx, count, n, row_bytes, (uint8_t*)dst);
  return check_launch();
}

int svdd_advance_rows(const void* src, const int32_t* slot, const int32_t* sel, int B, int M, int row_bytes, void* dst,
                      void* stream) {
  if (!src || !slot || !sel || !dst || B <= 0 || M <= 0 || row_bytes <= 0 || (row_bytes & 3)) return SVDD_E_ARG;
  hipLaunchKernelGGL(advance_rows_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, slot, sel,
                     B, M, row_bytes, (uint8_t*)dst);
  return check_launch();
}

static int launch_pos(void (*k)(PosArgs), const PosArgs& a, void* stream) {
  const int64_t N = (int64_t)a.R * a.L;
  hipLaunchKernelGGL(k, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch();
}

int svdd_x0hat(const float* logits, const uint8_t* xt, int R, int L, int layout, float* onehot_t, uint8_t* x0hat,
               void* stream) {
  if (!logits || !xt || (!onehot_t && !x0hat) || R <= 0 || L <= 0 || bad_layout(layout)) return SVDD_E_ARG;
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

int svdd_dps_probs(const float* logits, const uint8_t* x, int B, int L, float* probs4, void* stream) {
  if (!logits || !x || !probs4 || B <= 0 || L <= 0) return SVDD_E_ARG;
  const int64_t N = (int64_t)B * L;
  hipLaunchKernelGGL(dps_probs_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     DpsArgs{logits, x, nullptr, nullptr, probs4, nullptr, B, L, 0.0f, 0.0f, 0.0f});
  return check_launch();
}

int svdd_dps_probs_bwd(const float* logits, const uint8_t* x, const float* dprobs4, int B, int L, float* dlogits, float* direct,
                       void* stream) {
  if (!logits || !x || !dprobs4 || !dlogits || !direct || B <= 0 || L <= 0) return SVDD_E_ARG;
  const int64_t N = (int64_t)B * L;
  hipLaunchKernelGGL(dps_probs_bwd_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     DpsArgs{logits, x, dprobs4, nullptr, dlogits, direct, B, L, 0.0f, 0.0f, 0.0f});
  return check_launch();
}

int svdd_dps_guided_q(const float* logits, const uint8_t* x, const float* grad_backbone, const float* grad_direct, float dm, float mcs,
                      float scale, int B, int L, float* q, void* stream) {
  if (!logits || !x || !grad_backbone || !grad_direct || !q || B <= 0 || L <= 0) return SVDD_E_ARG;
  const int64_t N = (int64_t)B * L;
  hipLaunchKernelGGL(dps_guided_q_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     DpsArgs{logits, x, grad_backbone, grad_direct, q, nullptr, B, L, dm, mcs, scale});
  return check_launch();
}

int svdd_tds_resample(const float* reward_num, const float* reward_den, double alpha, const uint8_t* sample,
                      const double* u, int B, int L, uint8_t* x_next, int32_t* idx, double* work, void* stream) {
  if (!reward_num || !reward_den || !sample || !u || !x_next || !work || B <= 0 || L <= 0 || !(alpha != 0.0))
    return SVDD_E_ARG;
  TdsArgs a{reward_num, reward_den, alpha, sample, u, B, L, x_next, idx, work};
  TimedLaunch* t = timed_slot(8);              // one timed span over both launches: start of phase 1 .. stop of phase 2
  hipExtLaunchKernelGGL(tds_cdf_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, t ? t->start : nullptr, nullptr, 0, a);
  if (check_launch() != SVDD_OK) return SVDD_E_LAUNCH;
  hipExtLaunchKernelGGL(tds_gather_kernel, dim3((unsigned)(((int64_t)B + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                        nullptr, t ? t->stop : nullptr, 0, a);
  return check_launch();
}

int svdd_mt19937_uniform_f32(uint32_t* state, float* out, long long n, void* stream) {
  if (!state || !out || n < 0) return SVDD_E_ARG;
  if (n == 0) return SVDD_OK;
  TimedLaunch* t = timed_slot(9);
  hipExtLaunchKernelGGL(mt19937_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, t ? t->start : nullptr, t ? t->stop : nullptr, 0,
                        state, out, n);
  return check_launch();
}

}  // extern "C"
