/*
 * svdd_oracle.c — CPU ORACLE for the SVDD decode hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain scalar C restatement of the reference's per-step propose / score-select /
 * resample algorithm (masa-ue/SVDD, diffusion_gosai.py), used as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the
 * product path (svdd_amd/) may import, link or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * against golden vectors captured by running the reference's own Python on CPU in the
 * build container (tests/golden/make_golden.py writes the .npz fixtures under tests/golden).
 *
 * Arithmetic contract (mirrored by the HIP kernels):
 *  - every tensor op of the reference is one fp32 operation here, in the same order;
 *  - exp / log are evaluated CORRECTLY ROUNDED to fp32 (double libm, then one rounding).
 *    torch-CPU uses SLEEF u10 kernels (<=1 ulp); they agree with correct rounding on
 *    98.9 % (exp) / 99.9 % (log) of inputs [probed], so float outputs may differ from the
 *    reference by 1 ulp and token outputs only on measure-~1e-9 near-ties;
 *  - sums are sequential left-to-right in fp32 (matches torch-CPU logsumexp 100 % [probed]);
 *  - argmax returns the FIRST maximal index (torch.argmax on CPU).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no fast-math, no FMA contraction)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define V 5
#define MASK 4
#define NEG_INF_F (-1000000.0f) /* Diffusion.neg_infinity, diffusion_gosai.py:161 */

/* ------------------------------------------------------------------ math ---- */
static inline float expf_cr(float x) { return (float)exp((double)x); }
static inline float logf_cr(float x) { return (float)log((double)x); }

/* ---------------------------------------------------------------- mt19937 ---- */
/* std::mt19937 / numpy RandomState core. torch CPU: at::mt19937 (CPUGeneratorImpl). */
typedef struct { uint32_t mt[624]; int pos; } orc_mt_t;

void orc_mt_seed(orc_mt_t* s, uint32_t seed) {
  s->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->pos = 624;
}

static uint32_t mt_next(orc_mt_t* s) {
  if (s->pos >= 624) {
    uint32_t* mt = s->mt;
    for (int k = 0; k < 624; ++k) {
      uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
      mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->pos = 0;
  }
  uint32_t y = s->mt[s->pos++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* torch.manual_seed(seed); torch.rand(n) on CPU: one 32-bit draw per float, low 24 bits
 * scaled by 2^-24 (ATen uniform_real_distribution<float>; rand_like, diffusion_gosai.py:33). */
void orc_torch_rand_f32(orc_mt_t* s, float* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) out[i] = (float)(mt_next(s) & 0xFFFFFFu) * (1.0f / 16777216.0f);
}

/* np.random.seed(seed); np.random.random_sample(n): 53-bit doubles from two draws
 * (RandomState legacy_double; consumed by np.random.choice, diffusion_gosai.py:1282). */
void orc_numpy_random_sample(orc_mt_t* s, double* out, int64_t n) {
  for (int64_t i = 0; i < n; ++i) {
    uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
    out[i] = (a * 67108864.0 + b) / 9007199254740992.0;
  }
}

/* ----------------------------------------------------------------- philox ---- */
/* Philox4x32-10 (Salmon et al. 2011). Counter layout shared with the HIP kernels:
 *   ctr = { lo32(pos), hi32(pos), (step<<16)|m, stream }   key = { lo32(seed), hi32(seed) }
 *   pos = (row_offset + b) * L + l ; stream 0 -> the 5 uniforms of one draw: categories 0..3 take the
 *   top 24 bits of words 0..3, MASK the low bytes of words 0..2 ; stream 2 (pos = global row, m = 0)
 *   -> word 0 = select draw. */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

static inline float u24(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

void orc_philox_uniform5(uint64_t seed, uint64_t pos, uint32_t step, uint32_t m, float u[5]) {
  uint32_t c[4] = {(uint32_t)pos, (uint32_t)(pos >> 32), (step << 16) | m, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  u[0] = u24(c[0]); u[1] = u24(c[1]); u[2] = u24(c[2]); u[3] = u24(c[3]);
  uint32_t low = (c[0] & 0xFFu) | ((c[1] & 0xFFu) << 8) | ((c[2] & 0xFFu) << 16);
  u[4] = (float)low * (1.0f / 16777216.0f);
}

float orc_philox_select_uniform(uint64_t seed, uint64_t row, uint32_t step) {
  uint32_t c[4] = {(uint32_t)row, (uint32_t)(row >> 32), (step << 16), 2u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return u24(c[0]);
}

/* ------------------------------------------------------------------ layout ---- */
/* Memory layout of the [B,L,5] tensors (logits, q_xs, uniforms). The reference's CNN backbone
 * returns a permuted view (models/dnaconv.py:201 `feat.permute(0, 2, 1)`), so log_p_x0, q_xs and
 * the rand_like(q_xs) uniforms are all laid out [B][5][L] in memory; torch fills rand_like in
 * MEMORY order, so the mt19937 stream is consumed [b][v][l] for that backbone [probed] and
 * [b][l][v] for a backbone with contiguous output (e.g. DiT).
 *   layout 0 (BLV): (b,l,v) at (b*L + l)*5 + v     layout 1 (BVL): (b,l,v) at (b*5 + v)*L + l */
static inline int64_t at(int layout, int64_t b, int64_t l, int v, int64_t L) {
  return layout == 0 ? (b * L + l) * V + v : (b * V + v) * L + l;
}
static inline void load5(const float* p, int layout, int64_t b, int64_t l, int64_t L, float* z) {
  for (int v = 0; v < V; ++v) z[v] = p[at(layout, b, l, v, L)];
}
static inline void store5(float* p, int layout, int64_t b, int64_t l, int64_t L, const float* z) {
  for (int v = 0; v < V; ++v) p[at(layout, b, l, v, L)] = z[v];
}

/* --------------------------------------------------- SUBS parameterisation ---- */
/* Diffusion._subs_parameterization, diffusion_gosai.py:286-304, for one position.
 * z: raw backbone logits (5), xt: current token, out: log p(x0 | xt). */
static void subs_logp_1(const float* z, int xt, float* lp) {
  float zz[V];
  for (int v = 0; v < V; ++v) zz[v] = z[v];
  zz[MASK] = zz[MASK] + NEG_INF_F;                       /* :289 */
  /* torch.logsumexp: max, (x-max).exp().sum().log() + max           :293 */
  float mx = zz[0];
  for (int v = 1; v < V; ++v) if (zz[v] > mx) mx = zz[v];
  if (isinf(mx)) mx = 0.0f;                              /* ATen masked_fill_(abs==inf, 0) */
  float s = 0.0f;
  for (int v = 0; v < V; ++v) {
    float e = expf_cr(zz[v] - mx);
    s = (v == 0) ? e : s + e;
  }
  float lse = logf_cr(s) + mx;
  for (int v = 0; v < V; ++v) lp[v] = zz[v] - lse;
  if (xt != MASK) {                                      /* :300-303 */
    for (int v = 0; v < V; ++v) lp[v] = NEG_INF_F;
    lp[xt] = 0.0f;
  }
}

void orc_subs_logp(const float* logits, const uint8_t* x, int B, int L, int layout, float* logp) {
  for (int b = 0; b < B; ++b)
    for (int l = 0; l < L; ++l) {
      float z[V], lp[V];
      load5(logits, layout, b, l, L, z);
      subs_logp_1(z, x[(int64_t)b * L + l], lp);
      store5(logp, layout, b, l, L, lp);
    }
}

/* q_xs = exp(log_p_x0) * (mct - mcs); q_xs[..., MASK] = mcs   diffusion_gosai.py:1194-1196 */
static void qxs_1(const float* lp, float dm, float mcs, float* q) {
  for (int v = 0; v < V; ++v) q[v] = expf_cr(lp[v]) * dm;
  q[MASK] = mcs;
}

/* _sample_categorical for one position, diffusion_gosai.py:30-34:
 *   gumbel_norm = 1e-10 - (rand + 1e-10).log(); return (p / gumbel_norm).argmax(-1) */
static int sample_categorical_1(const float* q, const float* u) {
  int best = 0; float rbest = 0.0f;
  for (int v = 0; v < V; ++v) {
    float a = u[v] + 1e-10f;
    float g = 1e-10f - logf_cr(a);
    float r = q[v] / g;
    if (v == 0 || r > rbest) { rbest = r; best = v; }
  }
  return best;
}

void orc_sample_categorical(const float* q, const float* u, int64_t n, uint8_t* tok) {
  for (int64_t i = 0; i < n; ++i) tok[i] = (uint8_t)sample_categorical_1(q + i * V, u + i * V);
}

/* ------------------------------------------------------------------ propose ---- */
/* One SVDD propose phase, diffusion_gosai.py:1189-1208 (identical in :1387-1402):
 * SUBS log-probs -> q_xs -> M categorical draws merged with copy_flag -> one-hot(4).
 *  rng_kind 0: uniforms [M] x (layout) (torch-CPU stream order); 1: Philox(seed,row_offset,step).
 *  cand [B][M][L] u8 ; onehot [B*M][L][4] f32 ; q_xs (layout, nullable). */
void orc_propose(const float* logits, const uint8_t* x, float dm, float mcs,
                 int B, int L, int M, int layout, int rng_kind, const float* uniforms,
                 uint64_t seed, uint64_t row_offset, uint32_t step,
                 uint8_t* cand, float* onehot, float* q_xs) {
  for (int b = 0; b < B; ++b)
    for (int l = 0; l < L; ++l) {
      int64_t n = (int64_t)b * L + l;
      float z[V], lp[V], q[V];
      load5(logits, layout, b, l, L, z);
      subs_logp_1(z, x[n], lp);
      qxs_1(lp, dm, mcs, q);
      if (q_xs) store5(q_xs, layout, b, l, L, q);
      for (int m = 0; m < M; ++m) {
        float u[V];
        if (rng_kind == 0) load5(uniforms + (int64_t)m * B * L * V, layout, b, l, L, u);
        else orc_philox_uniform5(seed, (row_offset + (uint64_t)b) * (uint64_t)L + (uint64_t)l, step, (uint32_t)m, u);
        int t = sample_categorical_1(q, u);
        int c = (x[n] != MASK) ? x[n] : t;                 /* copy_flag merge, :1199,1203 */
        int64_t o = ((int64_t)b * M + m) * L + l;
        cand[o] = (uint8_t)c;
        float* oh = onehot + o * 4;                        /* transform_samples, :1462-1470 */
        for (int v = 0; v < 4; ++v) oh[v] = (c == v) ? 1.0f : 0.0f;
      }
    }
}

/* ------------------------------------------------------------------- select ---- */
/* scores [B][M] -> softmax(dim=1) -> argmax (or multinomial) -> gather row.
 * diffusion_gosai.py:1219-1227. softmax as ATen's CPU kernel: e=exp(s-max); r=1/sum(e); p=e*r.
 * mode 1 (multinomial, the commented-out :1223): inclusive fp32 cumsum of p, draw
 * u = Philox(seed,row,step,stream 2), pick first m with u*c_last < c_m. */
void orc_select(const float* scores, const uint8_t* cand, int B, int L, int M, int mode,
                uint64_t seed, uint64_t row_offset, uint32_t step,
                uint8_t* x_next, float* soft, int32_t* idx) {
  float* p = (float*)malloc(sizeof(float) * (size_t)M);
  for (int b = 0; b < B; ++b) {
    const float* s = scores + (int64_t)b * M;
    float mx = s[0];
    for (int m = 1; m < M; ++m) if (s[m] > mx) mx = s[m];
    float sum = 0.0f;
    for (int m = 0; m < M; ++m) { p[m] = expf_cr(s[m] - mx); sum = (m == 0) ? p[m] : sum + p[m]; }
    float r = 1.0f / sum;
    for (int m = 0; m < M; ++m) p[m] = p[m] * r;
    int best = 0;
    if (mode == 0) {
      for (int m = 1; m < M; ++m) if (p[m] > p[best]) best = m;
    } else {
      float u = orc_philox_select_uniform(seed, row_offset + (uint64_t)b, step);
      float c = 0.0f, tot = 0.0f;
      for (int m = 0; m < M; ++m) tot = (m == 0) ? p[m] : tot + p[m];
      float thr = u * tot;
      best = M - 1;
      for (int m = 0; m < M; ++m) {
        c = (m == 0) ? p[m] : c + p[m];
        if (thr < c) { best = m; break; }
      }
    }
    if (soft) memcpy(soft + (int64_t)b * M, p, sizeof(float) * (size_t)M);
    if (idx) idx[b] = best;
    memcpy(x_next + (int64_t)b * L, cand + ((int64_t)b * M + best) * L, (size_t)L);
  }
  free(p);
}

/* -------------------------------------------------------- x0hat / finalize ---- */
/* Tweedie candidate: argmax_v forward(xt)[...,v] over all 5 entries, one-hot(4), keep
 * unmasked tokens, transpose to [R][4][L].  diffusion_gosai.py:1415-1419,1430 (TDS :1263-1269). */
void orc_x0hat(const float* logits, const uint8_t* xt, int R, int L, int layout, float* onehot_t, uint8_t* x0hat) {
  for (int r = 0; r < R; ++r)
    for (int l = 0; l < L; ++l) {
      int64_t n = (int64_t)r * L + l;
      float z[V], lp[V];
      load5(logits, layout, r, l, L, z);
      subs_logp_1(z, xt[n], lp);
      int best = 0;
      for (int v = 1; v < V; ++v) if (lp[v] > lp[best]) best = v;
      int c = (xt[n] != MASK) ? xt[n] : best;
      if (x0hat) x0hat[n] = (uint8_t)c;
      for (int v = 0; v < 4; ++v) onehot_t[((int64_t)r * 4 + v) * L + l] = (c == v) ? 1.0f : 0.0f;
    }
}

/* Noise removal: x = forward(x, sigma)[:, :, :-1].argmax(-1)   diffusion_gosai.py:1049-1060 */
void orc_finalize(const float* logits, const uint8_t* x, int B, int L, int layout, int64_t* out_i64, uint8_t* out_u8) {
  for (int b = 0; b < B; ++b)
    for (int l = 0; l < L; ++l) {
      int64_t i = (int64_t)b * L + l;
      float z[V], lp[V];
      load5(logits, layout, b, l, L, z);
      subs_logp_1(z, x[i], lp);
      int best = 0;
      for (int v = 1; v < 4; ++v) if (lp[v] > lp[best]) best = v;
      if (out_i64) out_i64[i] = best;
      if (out_u8) out_u8[i] = (uint8_t)best;
    }
}

/* transform_samples, diffusion_gosai.py:1462-1470 == Enformer.py:269-277 */
void orc_transform_samples(const uint8_t* tok, int R, int L, int transposed, float* out) {
  for (int r = 0; r < R; ++r)
    for (int l = 0; l < L; ++l) {
      int c = tok[(int64_t)r * L + l];
      for (int v = 0; v < 4; ++v) {
        float f = (c == v) ? 1.0f : 0.0f;
        if (transposed) out[((int64_t)r * 4 + v) * L + l] = f;
        else out[((int64_t)r * L + l) * 4 + v] = f;
      }
    }
}

/* ------------------------------------------------------------- TDS resample ---- */
/* numpy's pairwise float32 sum (np.add.reduce on a contiguous float32 array), which is what
 * `ratio.sum()` runs at diffusion_gosai.py:1282. */
static float np_pairwise_sum_f32(const float* a, int64_t n) {
  if (n < 8) {
    float res = 0.0f;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  } else if (n <= 128) {
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  } else {
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
  }
}

/* diffusion_gosai.py:1280-1284:
 *   ratio = torch.exp(1.0/alpha * (reward_num - reward_den))       (fp32)
 *   p = ratio / ratio.sum()                                        (numpy float32, pairwise sum)
 *   idx = np.random.choice(B, B, p=p)  -> cdf = float64 cumsum(p); cdf /= cdf[-1];
 *                                         idx = searchsorted(cdf, u, side='right')
 *   return sample[idx]
 * u: the B doubles RandomState.random_sample(B) yields. */
void orc_tds_resample(const float* reward_num, const float* reward_den, double alpha,
                      const uint8_t* sample, const double* u, int B, int L,
                      uint8_t* x_next, int32_t* idx, float* ratio_out, double* cdf_out) {
  if (B <= 0) return;
  float* ratio = (float*)calloc((size_t)B, sizeof(float));
  double* cdf = (double*)calloc((size_t)B, sizeof(double));
  float inv_alpha = (float)(1.0 / alpha);          /* Python double 1.0/alpha, one rounding to fp32 (scalar * float tensor) */
  for (int b = 0; b < B; ++b) ratio[b] = expf_cr(inv_alpha * (reward_num[b] - reward_den[b]));
  float tot = np_pairwise_sum_f32(ratio, B);
  double c = 0.0;
  for (int b = 0; b < B; ++b) { c += (double)(ratio[b] / tot); cdf[b] = c; }
  double last = cdf[B - 1];
  for (int b = 0; b < B; ++b) cdf[b] /= last;
  for (int j = 0; j < B; ++j) {
    /* searchsorted(side='right'): number of cdf entries <= u */
    int lo = 0, hi = B;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cdf[mid] <= u[j]) lo = mid + 1; else hi = mid; }
    int k = lo < B ? lo : B - 1;
    if (idx) idx[j] = k;
    memcpy(x_next + (int64_t)j * L, sample + (int64_t)k * L, (size_t)L);
  }
  if (ratio_out) memcpy(ratio_out, ratio, sizeof(float) * (size_t)B);
  if (cdf_out) memcpy(cdf_out, cdf, sizeof(double) * (size_t)B);
  free(ratio); free(cdf);
}

/* ----------------------------------------------------------------- schedule ---- */
/* LogLinearNoise + move-chance prologue for one step (noise_schedule.py:144-145,
 * diffusion_gosai.py:1176-1187), correctly-rounded fp32:
 *   sigma = -log1p(-(1-1e-3)*t); move_chance = 1 - exp(-sigma).  out = {mct, mcs, mct-mcs}. */
void orc_move_chances(float t, float dt, float out[3]) {
  float c = (float)(1.0 - 1e-3);
  float ts = t - dt;
  float sig_t = -(float)log1p((double)(-(c * t)));
  float sig_s = -(float)log1p((double)(-(c * ts)));
  float mct = 1.0f - expf_cr(-sig_t);
  float mcs = 1.0f - expf_cr(-sig_s);
  out[0] = mct; out[1] = mcs; out[2] = mct - mcs;
}
