/*
 * svdd_hip.h — C ABI of libsvdd_hip.so, the MI355X (gfx950) implementation of the
 * SVDD per-step propose / score-select / resample hot path.
 *
 * The reference (masa-ue/SVDD) is pure Python and has no FFI layer; its boundary for
 * this path is the Python object protocol of `diffusion_gosai.Diffusion`. The entry
 * points below are what a binding for that path replaces, each citing the reference
 * lines whose tensor program it fuses (paths relative to the reference root).
 * INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless marked "host". Caller owns every buffer;
 *    nothing is allocated, freed or synchronised inside. Launches go to `stream`
 *    (a hipStream_t passed as void*; NULL = the legacy default stream). Every entry
 *    point is capturable in a hipGraph.
 *  - Tokens are uint8: 0..3 = A,C,G,T; 4 = MASK (diffusion_gosai.py:85,94-95). The
 *    reference's int64 token tensors carry the same values; the Python host mirror
 *    converts at the API edge.
 *  - Return value: 0 on success, a negative SVDD_E_* code otherwise. Launch failures
 *    are reported as SVDD_E_LAUNCH (query hipGetLastError for detail).
 *  - Arithmetic is fp32 in the reference's operation order; exp/log are evaluated
 *    correctly rounded to fp32 (see DESIGN.md "Arithmetic contract").
 */
#ifndef SVDD_HIP_H
#define SVDD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVDD_VOCAB 5      /* A,C,G,T,MASK */
#define SVDD_MASK 4
#define SVDD_MAX_M 1024   /* candidates per sample supported by svdd_select */

enum {
  SVDD_OK = 0,
  SVDD_E_ARG = -1,      /* null pointer / non-positive size / M out of range */
  SVDD_E_LAUNCH = -2,   /* hipLaunchKernel failed */
  SVDD_E_NODEVICE = -3  /* no HIP device / wrong architecture */
};

/* Memory layout of the [B,L,5] tensors (logits, q_xs, logp, replay uniforms).
 * The reference's CNN backbone returns `feat.permute(0, 2, 1)` (models/dnaconv.py:201), a view of a
 * [B,5,L] buffer; log_p_x0, q_xs and rand_like(q_xs) inherit those strides and torch fills
 * rand_like in MEMORY order. A drop-in must therefore accept both images. */
enum {
  SVDD_LAYOUT_BLV = 0,  /* element (b,l,v) at (b*L + l)*5 + v   — contiguous [B,L,5] (e.g. DiT) */
  SVDD_LAYOUT_BVL = 1   /* element (b,l,v) at (b*5 + v)*L + l   — [B,5,L], the conv-native image */
};

/* Uniform source for the categorical draws. */
enum {
  SVDD_RNG_REPLAY = 0,  /* uniforms supplied by the caller (bit-exact replay of the
                           reference's torch-CPU mt19937 stream, diffusion_gosai.py:33) */
  SVDD_RNG_PHILOX = 1   /* counter-based Philox4x32-10 generated in-kernel; keyed by
                           (seed, step, global row, candidate, position) so output is
                           invariant to how rows are sharded over GPUs */
};

typedef struct svdd_rng {
  int32_t kind;            /* SVDD_RNG_REPLAY | SVDD_RNG_PHILOX */
  uint32_t step;           /* PHILOX: diffusion step index (0..S-1) */
  const float* uniforms;   /* REPLAY: M consecutive blocks of B*L*5 fp32 in [0,1), each laid out as
                              `uniforms_layout` — the order M rand_like(q_xs) calls consume, i.e. the
                              memory order of the REFERENCE backbone's output (BVL for its CNN) */
  uint64_t seed;           /* PHILOX: 64-bit key */
  uint64_t row_offset;     /* PHILOX: global index of this shard's row 0 ; REPLAY with uniforms_rows > 0: likewise */
  int32_t uniforms_layout; /* REPLAY: SVDD_LAYOUT_* of each uniforms block (independent of `layout`) */
  int32_t uniforms_rows;   /* REPLAY: 0 = each block holds B rows (this batch's own uniforms); > 0 = each block holds the uniforms of
                              a WHOLE batch of that many rows, of which this call's B rows start at `row_offset` — a rank of a
                              batch-sharded decode replays the global stream and reads its slice */
} svdd_rng_t;

/* Selection rule of svdd_select. */
enum {
  SVDD_SELECT_ARGMAX = 0,      /* reference behaviour: argmax(softmax(scores)) (diffusion_gosai.py:1220,1225) */
  SVDD_SELECT_MULTINOMIAL = 1  /* the commented-out torch.multinomial variant (:1223); PHILOX only */
};

/*
 * svdd_propose — replaces, per diffusion step:
 *   Diffusion._subs_parameterization               diffusion_gosai.py:286-304
 *   q_xs construction                               diffusion_gosai.py:1194-1196
 *   copy_flag                                       diffusion_gosai.py:1199
 *   M x _sample_categorical + copy-flag merge       diffusion_gosai.py:30-34, 1203
 *   M x transform_samples(...).float()              diffusion_gosai.py:1462-1470, 1208
 *
 *  logits  [B,L,5] fp32  raw backbone output in `layout` (NOT modified; the reference edits it in place)
 *  x       [B,L]   u8    current tokens x_t
 *  dm      = fl32(move_chance_t - move_chance_s),  mcs = move_chance_s   (:1184-1187)
 *  cand    [B,M,L] u8    out: the M proposals per sample
 *  onehot  [B*M,L,4] fp32 out: value-net input, row (b*M+m); MASK rows all-zero
 *  q_xs    [B,L,5] fp32  out in `layout`, may be NULL (only the per-step API returns it, :1228)
 */
int svdd_propose(const float* logits, const uint8_t* x, float dm, float mcs,
                 int B, int L, int M, int layout, const svdd_rng_t* rng,
                 uint8_t* cand, float* onehot, float* q_xs, void* stream);

/*
 * svdd_sample_categorical — M x (_sample_categorical(q) merged with copy_flag) + one-hot, for a caller-built
 * q (e.g. the DPS baseline's guided q_xs * exp(guidance), diffusion_gosai.py:1311-1318; :30-34).
 *  q [B,L,5] fp32 in `layout`, all entries >= 0; other arguments as svdd_propose.
 */
int svdd_sample_categorical(const float* q, const uint8_t* x, int B, int L, int M, int layout,
                            const svdd_rng_t* rng, uint8_t* cand, float* onehot, void* stream);

/*
 * svdd_select — replaces torch.stack(scores,1) -> softmax(dim=1) -> argmax(dim=1) ->
 * per-row Python gather + stack                     diffusion_gosai.py:1219-1227 (= :1451-1459)
 *
 *  scores [B,M] fp32 (row b, candidate m) ; cand [B,M,L] u8
 *  x_next [B,L] u8 out ; soft [B,M] fp32 out (softmax, may be NULL) ; idx [B] i32 out (may be NULL)
 *  (x_next == NULL with idx != NULL, M <= 64: the decision only, no row gather — round 6's measurement of what the gather costs)
 *  rng is read only for SVDD_SELECT_MULTINOMIAL (must be PHILOX).
 */
int svdd_select(const float* scores, const uint8_t* cand, int B, int L, int M, int mode,
                const svdd_rng_t* rng, uint8_t* x_next, float* soft, int32_t* idx,
                void* stream);

/*
 * svdd_x0hat — SVDD-PM / TDS posterior-mean candidate (Tweedie):
 *   forward()'s _subs_parameterization + argmax(dim=2) + one_hot + keep-unmasked merge +
 *   .float().transpose(1,2)                         diffusion_gosai.py:1415-1419,1430 ; :1263-1269
 *  logits [R,L,5] raw backbone output for tokens xt [R,L] ; out onehot_t [R,4,L] fp32 (may be NULL when x0hat is given) ;
 *  x0hat [R,L] u8 out (may be NULL).
 */
int svdd_x0hat(const float* logits, const uint8_t* xt, int R, int L, int layout,
               float* onehot_t, uint8_t* x0hat, void* stream);

/*
 * svdd_finalize — noise-removal step: forward() then logits[:,:,:-1].argmax(-1)
 *                                                   diffusion_gosai.py:1049-1060
 *  out_i64 [B,L] int64 (the API's LongTensor) and/or out_u8 [B,L]; either may be NULL.
 */
int svdd_finalize(const float* logits, const uint8_t* x, int B, int L, int layout,
                  int64_t* out_i64, uint8_t* out_u8, void* stream);

/*
 * svdd_transform_samples — tokens -> one-hot(4) with MASK rows zero
 *                                                   diffusion_gosai.py:1462-1470 ≡ Enformer.py:269-277
 *  transposed == 0: out [R,L,4] (value-net layout) ; != 0: out [R,4,L] (reward-model layout, Enformer.py:447)
 */
int svdd_transform_samples(const uint8_t* tok, int R, int L, int transposed, float* out,
                           void* stream);

/*
 * svdd_subs_logp — Diffusion.forward()'s SUBS re-parameterisation alone
 *                                                   diffusion_gosai.py:286-304
 *  logp [B,L,5] out. (Used by the per-step API mirror and the DPS baseline.)
 */
int svdd_subs_logp(const float* logits, const uint8_t* x, int B, int L, int layout, float* logp,
                   void* stream);

/*
 * Exact work-skipping. A candidate that unmasked nothing is a copy of its parent x_t
 * (diffusion_gosai.py:1203), and with time_conditioning off (:334-335) every net output for it equals the parent's; a
 * row whose selected candidate is such a copy keeps its logits. The helpers below let the engine run the nets only on
 * the LIVE candidates / rows without a host round trip: the compacted index list and its length stay in device memory
 * and the net kernels take the length as a device pointer (`count` arguments below).
 *
 * svdd_compact_flags — stable compaction of n flags (one workgroup): flags[i] != 0 -> live_idx[k] = i, slot[i] = k with
 *   k = number of live items before i; else slot[i] = -1; count[0] = number of live items.
 * svdd_compact_by_key — the same compaction ORDERED by key, largest first (stable inside a key; keys clamp to 15, key <= 0 = not
 *   live): live_idx lists the live items by descending key, slot[i] = position of item i in it or -1. With key = row tiles of a
 *   candidate's window (the flags svdd_candidate_windows writes) the windowed tower's long workgroups are dispatched first.
 *   split > 0: count must hold 3 ints; count[1] = min(count[0], split), count[2] = max(count[0] - split, 0) — the lengths of the
 *   list's two parts [0, split) and [split, ...), for callers that run the parts as separate launches (fused.py: the GRU's second round).
 * svdd_gather_rows   — dst[i,:] = src[idx[i],:] for i < count[0] (count may be NULL = n); rows of row_bytes bytes.
 * svdd_advance_rows  — the selected candidate becomes the next parent: for every b, if slot[b*M + sel[b]] >= 0 then
 *   dst[b,:] = src[slot[...],:], else dst[b] is left untouched; rows of row_bytes (multiple of 4).
 * svdd_select_compact — svdd_select on compacted scores: candidate (b, m) has score scores[slot[b*M+m]] if its slot is
 *   >= 0, else parent_score[b]. Additionally writes sel_score[b] (the selected candidate's score = the NEXT step's
 *   parent score) and changed[b] (1 iff x_next[b] differs from x[b]). slot == NULL: exactly svdd_select.
 */
int svdd_compact_flags(const int32_t* flags, int n, int32_t* live_idx, int32_t* slot, int32_t* count, void* stream);
int svdd_compact_by_key(const int32_t* key, int n, int32_t* live_idx, int32_t* slot, int32_t* count, int split, void* stream);
int svdd_gather_rows(const void* src, const int32_t* idx, const int32_t* count, int n, int row_bytes, void* dst, void* stream);
int svdd_advance_rows(const void* src, const int32_t* slot, const int32_t* sel, int B, int M, int row_bytes, void* dst,
                      void* stream);
int svdd_select_compact(const float* scores, const int32_t* slot, const float* parent_score, const uint8_t* cand, int B,
                        int L, int M, int mode, const svdd_rng_t* rng, uint8_t* x_next, float* soft, int32_t* idx,
                        float* sel_score, int32_t* changed, void* stream);

/*
 * svdd_tds_resample — SMC/TDS baseline resampling step
 *   ratio = exp(fl32(1.0/alpha) * (num-den)); p = ratio/ratio.sum(); idx = np.random.choice(B,B,p=p);
 *   (alpha is the caller's Python float, i.e. a double: the reference rounds the double quotient 1.0/alpha to fp32 once)
 *   return sample[idx]                              diffusion_gosai.py:1280-1284
 *  reward_num, reward_den [B] fp32 ; sample [B,L] u8 ; u [B] fp64 uniforms in [0,1)
 *  (REPLAY of numpy's RandomState.random_sample, supplied by the host) ;
 *  out x_next [B,L] u8 ; idx [B] i32 (may be NULL) ; work [2*B] fp64 scratch (cdf + ratio).
 *  Two launches on `stream`: the CDF (one workgroup; the float64 cumsum is a serial chain by numpy's definition) and the
 *  searchsorted + row gather (whole chip). Any B; the pairwise sum's blocks run in parallel up to B = 131072.
 */
int svdd_tds_resample(const float* reward_num, const float* reward_den, double alpha,
                      const uint8_t* sample, const double* u, int B, int L,
                      uint8_t* x_next, int32_t* idx, double* work, void* stream);

/*
 * svdd_mt19937_uniform_f32 — torch's CPU generator stream on the device (parity-mode RNG without host traffic)
 *   replaces torch.rand / rand_like on the global CPU generator: diffusion_gosai.py:33 (`torch.rand_like(categorical_probs)`),
 *   i.e. at::mt19937 (std::mt19937) + uniform_real_distribution<float>: out[i] = (y_i & 0xFFFFFF) * 2^-24, y_i the i-th
 *   tempered 32-bit output, strictly in sequence.
 *  state [625] u32 in device memory: the 624 state words + the index of the next output (624 = twist first, as
 *  torch.manual_seed leaves it); advanced in place by n outputs (any rotation of the word window is a valid state: the
 *  recurrence is shift-invariant; svdd_amd/ops.py converts from / to torch.get_rng_state()'s layout).
 *  out [n] fp32. One workgroup (the recurrence is serial): ~0.3 ns per output; launch it on a side stream a step ahead.
 */
int svdd_mt19937_uniform_f32(uint32_t* state, float* out, long long n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Net kernels — internals of the value network (reference Enformer.py), not of the sampler. They are
 * optional accelerations of PyTorch modules (svdd_amd/fused.py); the sampler API above never needs them.
 *
 * svdd_gru_bidir_f32 — bidirectional single-layer GRU, input = hidden = 64, fp32
 *   (reference Enformer.py:1595-1602 nn.GRU(64, 64, bidirectional=True, batch_first=True), gate order r,z,n).
 *   x [n,L,64] ; out [2,n,L,64] = per-direction hidden states (the caller sums them, Enformer.py:1617).
 *   wpack [2][4][64][96], bpack [2][4][64]: weights repacked per MFMA lane, see svdd_amd/fused.py:pack_gru.
 *   `count` arguments of the net kernels: device scalar with the number of VALID rows of a compacted batch (the tensors
 *   stay laid out for n rows); NULL = n. See "Exact work-skipping" above. */
int svdd_gru_bidir_f32(const float* x, const float* wpack, const float* bpack, float* out, int n, int L,
                       const int32_t* count, void* stream);

/* The same GRU with a backward pass to its INPUT (weights are frozen in every decode path) — for the gradient-guidance baseline
 * of BASELINE.json configs[4] (reference diffusion_gosai.py:1321-1330 compute_gradient_DPS differentiates the reward model, whose
 * trunk holds this GRU, Enformer.py:1595-1602, with respect to the one-hot input). csrc/svdd_gru_train.hip.
 * svdd_gru_bidir_train_f32: forward as svdd_gru_bidir_f32 (same bits; no compaction), also writing
 *   save [2 dirs][n][L][4][64] = r, z, n, W_hn h + b_hn per step.
 * svdd_gru_bidir_bwd_f32: grad_out [2][n][L][64] (gradient of the per-direction outputs), out / save of the forward call,
 *   wpack_bwd [2 dirs][4 waves][64 lanes][96] packed by svdd_amd.fused.pack_gru_bwd -> dx [2][n][L][64], each direction's
 *   contribution to d loss / d x (the caller adds the two). */
/* The element-wise half of a dilated-CNN backbone layer under autograd (reference models/dnaconv.py:212-247 CNNModel.forward2:
 * feat' = relu(conv(LayerNorm(feat + time_bias)) + b) + feat; differentiated with respect to its input by the DPS baseline,
 * diffusion_gosai.py:1321-1330). Rows [rows, channels] fp32 channels-last, channels in {64, 128, 256}; tb [rows / rows_per_seq,
 * channels] = the time bias per sequence. Weights are frozen: input gradients only.
 * svdd_bb_layer_fwd_f32: f_out = relu(y + bias) + f_prev, mask = (y + bias > 0) (y NULL: f_out = f_prev, nothing written), then
 *   hn = LayerNorm(f_out + tb) gamma + beta (gamma NULL: none).
 * svdd_bb_layer_bwd_f32: g_out = g_in + dLayerNorm(g_hn) at h = f_in + tb ; gt_out = g_out where mask_prev else 0 (both NULL: none). */
int svdd_bb_layer_fwd_f32(const float* y, const float* bias, const float* f_prev, const float* tb, const float* gamma,
                          const float* beta, float eps, float* f_out, uint8_t* mask, float* hn, int64_t rows, int rows_per_seq,
                          int channels, void* stream);
int svdd_bb_layer_bwd_f32(const float* g_hn, const float* f_in, const float* tb, const float* gamma, float eps, const float* g_in,
                          const uint8_t* mask_prev, float* g_out, float* gt_out, int64_t rows, int rows_per_seq, int channels,
                          void* stream);
int svdd_gru_bidir_train_f32(const float* x, const float* wpack, const float* bpack, float* out, float* save, int n, int L,
                             void* stream);
int svdd_gru_bidir_bwd_f32(const float* grad_out, const float* out, const float* save, const float* wpack_bwd, float* dx, int n,
                           int L, void* stream);
/* svdd_value_tail_f32 — everything of the ConvGRU value net after the GRU, in one pass over the two GRU outputs:
 *   out[n][t] = b_eff[t] + mean_l sum_c w_eff[c][t] * relu(b1'[c] + sum_k W1'[c][k] * norm(h_fwd + h_bwd)[n][l][k])
 *   norm = LayerNorm over the 64 channels WITHOUT affine (eps 1e-5); the caller folds the LayerNorm affine into the
 *   linear map: W1' = W1 diag(gamma), b1' = b1 + W1 beta
 *   (reference Enformer.py:1617 direction sum; :2010-2047 FeedForwardBlock; :2131-2173 ConvHead with pool "avg").
 *   h_fwd, h_bwd [n,L,64] ; w1pack [64 lanes][128]: lane (j = lane & 15, g = lane >> 4) holds
 *   W1'[16 ct + j][16 (s / 4) + 4 g + s % 4] at 16 ct + s (svdd_amd/fused.py:pack_tail) ; b1 = b1' [128] ;
 *   w_eff [128][n_tasks] = (W_head W_2)^T, b_eff [n_tasks] = W_head b_2 + b_head ; out [n][n_tasks] ; n_tasks <= 4. */
int svdd_value_tail_f32(const float* h_fwd, const float* h_bwd, const float* w1pack, const float* b1,
                        const float* w_eff, const float* b_eff, float* out, int n, int L, int n_tasks,
                        const int32_t* count, void* stream);
/* tests / experiments: 2 selects the both-directions-per-workgroup scheduling (balanced but measured slower), else default */
int svdd_gru_set_mode(int mode);

/* svdd_epilogue_ln_f32 — fused convolution epilogue (+ next layer's LayerNorm) on channels-last rows [rows, C],
 *   C in {64,128,256}: t = y + bias ; f_out = relu(t) + f_prev (act 0) | relu(t + f_prev) (act 1) | t + f_prev (act 2);
 *   hn = LayerNorm(f_out + tb) * gamma + beta (eps 1e-5).  bias, f_prev, tb may be NULL; hn NULL skips the norm;
 *   f_out NULL skips storing the pre-norm sum.
 *   Replaces the bias/ReLU/residual/LayerNorm tensor ops between two convolutions of the dilated-CNN backbone
 *   (reference models/dnaconv.py:188-197) and of the value net's conv tower (Enformer.py:2269-2285). */
int svdd_epilogue_ln_f32(const float* y, const float* bias, const float* f_prev, const float* tb,
                         const float* gamma, const float* beta, float* f_out, float* hn, int64_t rows,
                         int channels, int act, void* stream);

/* svdd_conv1d_cl_f32 — dilated Conv1d with "same" zero padding on channels-last activations, fp32 on the
 *   exact-fp32 matrix cores, NO bias (the epilogue kernel adds it):
 *     y[n,l,co] = sum_{t,ci} x[n, l + (t - taps/2)*dilation, ci] * W[co,ci,t]
 *   x [n,L,cin], y [n,L,cout], cin/cout in {64,128}, taps odd, L <= 224.
 *   wpack [taps][cin/32][cout][32] = W[co][32c + k][t]   (svdd_amd/fused.py:pack_conv).
 *   Replaces the nn.Conv1d calls of reference models/dnaconv.py:151-156,196 and Enformer.py:2245-2253,2271. */
int svdd_conv1d_cl_f32(const float* x, const float* wpack, float* y, int n, int L, int cin, int cout,
                       int taps, int dilation, const float* bias, const float* f_prev, int act,
                       const float* tb, const float* gamma, const float* beta, float* hn, void* stream);
/*   fused epilogue (specialised shapes only: 128->128 x 9 taps x dilation {1,4,16,64}, 64->64 x 5 taps; L in {200,50}):
 *   act -1: y = conv ; 0: y = relu(conv + bias) + f_prev ; 1: y = relu(conv + bias + f_prev) ; 2: y = conv + bias + f_prev
 *   (bias, f_prev may be NULL); with hn != NULL additionally hn = LayerNorm(y + tb) * gamma + beta (eps 1e-5), the
 *   next layer's normalised input (tb may be NULL).  Other shapes take the generic kernel and require act = -1. */
/* tests only: != 0 forces the dynamically scheduled kernel instead of the per-(dilation,L) specialisations */
int svdd_conv1d_set_dynamic(int on);

/* svdd_conv_tower_f32 — the whole conv tower of the ConvGRU value net in one launch, activations resident in LDS:
 *   a0 = relu(conv15(onehot) + b0) ; a_{k+1} = relu(conv5(a_k) + b_k [+ a_k])  (k < nlayers, 64 channels, dilation 1)
 *   (reference Enformer.py:1634-1751: Stem + ConvBlocks "CDNRA"; eval-mode BatchNorm folded by the caller).
 *   onehot [n,L,4] ; tiles [2 + 10*nlayers][64][32] weight tiles in execution order (svdd_amd/fused.py:pack_tower) ;
 *   bias [1 + nlayers][64] ; out [n,L,64] ; residual_mask bit k = layer k adds its input ; L <= 208. */
int svdd_conv_tower_f32(const float* onehot, const float* tiles, const float* bias, float* out, int n, int L,
                        int nlayers, int residual_mask, const int32_t* count, void* stream);

/* tests / A-B: 1 = the backbone kernels always fill their 208-row tile with whole sequences (round-1 behaviour); 0 (default)
 * = when several sequences fit a tile (L <= 104) the number taken minimises rounds x tile cost, chosen on the host or, for a
 * device-side row count, by the workgroups themselves (csrc/svdd_spt.h). A row's logits do not depend on the choice. */
int svdd_set_backbone_packing(int full);

/* tests / A-B of the fp32 tower kernels (all produce the same bits): 1 = first generation (runtime tile predicates),
 * 2 / 3 = second generation (live-tile count as a template parameter) with two / one column tile per wave; 0 = default
 * (= 3, the fastest on whole sequences and on windows: profiles/r02_tower_ab.txt) */
int svdd_set_tower_version(int v);

/* svdd_candidate_windows + svdd_conv_tower_windows_f32 — the conv tower on the M candidates of every sample, sharing
 *   the work they have in common with their parent x_t (SVDD-MC scoring, reference diffusion_gosai.py:1203-1209: the
 *   candidates are copies of x_t with a few MASKs replaced). The tower's receptive field is +-17 rows, so a
 *   candidate's tower output differs from its parent's only near the replaced positions; only a row window around
 *   them is computed, the other rows are copied from the parent's output. Bit-identical to svdd_conv_tower_f32 on
 *   the candidates (tests/test_fused_gpu.py::test_tower_windows_equal_full_tower).
 *   svdd_candidate_windows: cand [B,M,L] u8, x [B,L] u8 -> win [B*M][2] = (w0, w1), multiples of 16 covering the
 *     positions where the candidate differs from its parent +- margin (27 for the 5-layer tower); (0, 0) if none.
 *   svdd_conv_tower_windows_f32: onehot [n = B*M, L, 4] (row b*M + m), win from above, parent_out [B, L, 64] =
 *     svdd_conv_tower_f32 of the parents' one-hot; out [n, L, 64]. 104 < L <= 208, nlayers = 5.
 *   flags [B*M] (may be NULL): the number of row tiles of the window ((w1 - w0) / 16 >= 1) if the candidate differs from its
 *     parent, 0 for an exact copy (input of svdd_compact_flags / key of svdd_compact_by_key). live_idx / count (may be NULL): process only the listed candidates; workgroup i handles
 *     candidate live_idx[i] and writes rows [i*L, (i+1)*L) of out (a compacted batch). */
int svdd_candidate_windows(const uint8_t* cand, const uint8_t* x, int B, int L, int M, int margin, int32_t* win,
                           int32_t* flags, void* stream);
int svdd_conv_tower_windows_f32(const float* onehot, const float* tiles, const float* bias, const int32_t* win,
                                const float* parent_out, float* out, int n, int L, int M, int nlayers,
                                int residual_mask, const int32_t* live_idx, const int32_t* count, void* stream);

/* svdd_backbone_cnn_f32 — the whole dilated-CNN masked-diffusion backbone at sigma = 0 in ONE launch
 *   (reference models/dnaconv.py:176-210 as called from diffusion_gosai.py:334-340): one-hot + 9-tap first conv,
 *   nlayers x [LayerNorm(f + tb_i) -> dilated 9-tap conv 128->128 -> ReLU -> + f], then the two 1x1 convs of
 *   final_conv. The residual stream stays in registers, the normalised activations in LDS; nothing but the tokens
 *   and the logits touches HBM.  hidden_dim = 128, alphabet 5, L <= 208, nlayers <= 32.
 *   x [n,L] u8 tokens ; table0 [9][5][128] = W_first[co][c][t] ; tiles [nlayers][4][9][128][32] = W_i[co][32c+k][t]
 *   followed by [4][128][32] = W_f1[co][32c+k] ; vec [nlayers+2][4][128]: row 0 = {b_first}, row 1+i = {b_i, tb_i,
 *   gamma_i, beta_i}, row nlayers+1 = {b_f1} ; w2 [5][128] then b2 [5] ; dilations: HOST int[nlayers] ;
 *   out [n,L,5] raw logits (layout BLV).  Packing: svdd_amd/fused.py:pack_backbone.
 *   Exact work-skipping: count (device scalar, NULL = n) limits the forward to the first count[0] compact rows; row_idx
 *   (NULL = identity) makes compact row r read the tokens of sequence row_idx[r] of x, and with out_scatter != 0 write
 *   its logits to row row_idx[r] of out (per-row logits cache: only the rows that changed are recomputed, in place)
 *   instead of row r (candidate compaction). A row's logits are the same bits wherever it is evaluated. */
int svdd_backbone_cnn_f32(const uint8_t* x, const float* table0, const float* tiles, const float* vec,
                          const float* w2, float* out, int n, int L, int nlayers, const int* dilations,
                          const int32_t* count, const int32_t* row_idx, int out_scatter, void* stream);
/* svdd_backbone_cnn_save_f32 + svdd_backbone_cnn_grad_f32 — the backbone's input gradient, one launch each way (the gradient-
 *   guidance baseline DPS: reference diffusion_gosai.py:1321-1330 differentiates reward(softmax(E[x0 | x_t])) with respect to
 *   onehot(x_t) through models/dnaconv.py:212-247; the weights are frozen). 104 < L <= 208 (one sequence per workgroup).
 *   svdd_backbone_cnn_save_f32: svdd_backbone_cnn_f32 on the tokens x (the SAME BITS in `out`) that also writes, in the kernel's
 *     lane-private layout, xhat [n][nlayers][56][512] f32 (LayerNorm'd value before the affine map), rstd [n][nlayers][208] f32
 *     and mask [n][nlayers + 2][512] u64 (ReLU decisions of the first layer, every conv layer and final_conv's first 1x1).
 *   svdd_backbone_cnn_grad_f32: dlogits [n,L,5] = d loss / d `out` -> dx [n,L,5] = d loss / d onehot(x), through the transposed
 *     1x1 convs, 20 x [ReLU', transposed dilated conv (the same implicit GEMM on tiles_bwd), LayerNorm backward from xhat / rstd,
 *     residual] and the first conv's transpose. tiles_bwd: W_f1^T as [4][128][32], then layers nlayers-1 .. 0 as [4][9][128][32] of
 *     W'[ci][32c+k][t] = W[32c+k][ci][8-t] ; gamma [nlayers][128] ; w2, table0 as in svdd_backbone_cnn_f32.
 *     Packing: svdd_amd/fused.py:pack_backbone_grad. */
int svdd_backbone_cnn_save_f32(const uint8_t* x, const float* table0, const float* tiles, const float* vec, const float* w2,
                               float* out, int n, int L, int nlayers, const int* dilations, float* xhat, float* rstd,
                               unsigned long long* mask, void* stream);
int svdd_backbone_cnn_grad_f32(const float* dlogits, const float* tiles_bwd, const float* gamma, const float* w2,
                               const float* table0, const float* xhat, const float* rstd, const unsigned long long* mask,
                               float* dx, int n, int L, int nlayers, const int* dilations, void* stream);
/* svdd_backbone_set_workspace — caller-owned scratch for the small-batch form of svdd_backbone_cnn_f32 (several workgroups per
 * sequence exchange the LayerNorm'd image of every layer through it): ws = device memory of `bytes` >= n_max * (2 * 208 * 128 * 4
 * + 4) + 4 bytes for batches of up to n_max sequences (n_max = 128 covers every case the split is used for); NULL: none (one
 * workgroup per sequence at every batch size). Launches that use it must not overlap. svdd_backbone_split_status: *err = 1 if a
 * group barrier ever timed out (never in a correct launch: the launcher only splits when every workgroup is resident). */
int svdd_backbone_set_workspace(void* ws, long long bytes);
int svdd_backbone_split_status(int* err);

/* ------------------------------------------------------------------------------------------------
 * Split-precision net kernels (svdd_amd/csrc/svdd_lp_*.hip) — the same functions as the *_f32 net kernels above
 * with the matrix products on the 16-bit matrix cores (fp32 accumulate). An explicit opt-in of the caller
 * (Diffusion.precision); the exact-fp32 kernels remain the default and the parity reference.
 *   SVDD_PREC_F16X3 / BF16X3: fp32 operands split hi + lo on the fly, a*b = ahi*bhi + ahi*blo + alo*bhi (3 MFMAs):
 *     f16x3 is fp32-class (a 22-bit operand: 1.5e-7 of sum|a b| at K = 1152 vs 1.8e-7 for the fp32 MFMA chain); bf16x3 is NOT — its
 *     operand keeps 16 bits (5.3e-7 of sum|a b| per product, backbone logits 4.5e-5 from an fp64 forward: 10 x fp32's 4.5e-6);
 *   SVDD_PREC_F16 / BF16: one pass on the 16-bit roundings of the operands (~3e-5 / ~3e-4 of sum|a b|).
 * Everything that is not a matrix product stays fp32. */
enum { SVDD_PREC_F32 = 0, SVDD_PREC_F16X3 = 1, SVDD_PREC_BF16X3 = 2, SVDD_PREC_F16 = 3, SVDD_PREC_BF16 = 4 };

/* svdd_backbone_cnn_lp — svdd_backbone_cnn_f32 on the 16-bit matrix cores. x, table0, vec, w2, out, dilations as
 *   there. tiles: [nlayers*36 + 4] weight tiles in the order (layer, chunk, tap) then the 4 chunks of W_f1, each
 *   [4 cg][64 lanes][2 ct][P][8] 16-bit values, P = 2 (hi, lo) for the x3 modes and 1 otherwise: lane (j = lane & 15,
 *   g = lane >> 4) of column group cg holds s_w * W[32 cg + 2 j + ct][32 c + 8 g + e][t], e = 0..7.
 *   lscale [nlayers + 1][2] = {sa, 1 / (sa * s_w)}: power-of-two scales of the activations / weights of each stage
 *   (1 for bf16).  Packing: svdd_amd/fused.py:pack_backbone_lp. */
int svdd_backbone_cnn_lp(const uint8_t* x, const float* table0, const void* tiles, const float* vec,
                         const float* lscale, const float* w2, float* out, int n, int L, int nlayers,
                         const int* dilations, int prec, const int32_t* count, const int32_t* row_idx, int out_scatter,
                         void* stream);

/* svdd_conv_tower_lp / svdd_conv_tower_windows_lp — svdd_conv_tower_f32 / svdd_conv_tower_windows_f32 on the 16-bit
 *   matrix cores. The input is the TOKEN tensor (u8, 4 = MASK -> zero one-hot row; reference transform_samples,
 *   diffusion_gosai.py:1462-1470), not the fp32 one-hot: tok [n,L] (windows: the candidates [B,M,L] flattened).
 *   tiles: [2 + 10*nlayers] weight tiles in execution order (stem chunks 0,1 with k = 4*tap + channel, then per layer,
 *   per 32-channel chunk, per tap), each [2 cp][64 lanes][2 ct][P][8] 16-bit: lane (j, g) of column pair cp holds
 *   s_w * W[32 cp + 2 j + ct][8 g + e]; inv [1 + nlayers] = 1 / s_w of each stage; bias as in the fp32 kernel.
 *   out (and parent_out): [n, L, P, 64] 16-bit planes — per row the hi halves of the 64 channels, then (x3 modes) the
 *   lo halves: the form svdd_gru_bidir_lp consumes directly (value = hi + lo).
 *   windows: live_idx [count] (may be NULL = identity) lists the candidates to process, `count` (device scalar, may be
 *   NULL = n) how many — workgroup i handles candidate live_idx[i] and writes rows [i*L, (i+1)*L) of `out`, so a
 *   compacted batch needs no host round trip (exact work-skipping). Packing: svdd_amd/fused.py:pack_tower_lp. */
int svdd_conv_tower_lp(const uint8_t* tok, const void* tiles, const float* bias, const float* inv, void* out,
                       int n, int L, int nlayers, int residual_mask, const int32_t* count, int prec, void* stream);
int svdd_conv_tower_windows_lp(const uint8_t* cand, const void* tiles, const float* bias, const float* inv,
                               const int32_t* win, const void* parent_out, void* out, int n, int L, int M,
                               int nlayers, int residual_mask, const int32_t* live_idx, const int32_t* count,
                               int prec, void* stream);

/* svdd_gru_bidir_lp — svdd_gru_bidir_f32 on the 16-bit matrix cores. wpack [2 dirs][4 waves][64 lanes][6][2][P][8]
 *   16-bit: lane (j, g) of wave w holds s_w * W_m[16 w + j][32 c + 8 g + e] for m = ir, hr, iz, hz, in, hn and chunk
 *   c = 0, 1; bpack as in the fp32 kernel; inv [2] = 1 / s_w per direction. `count` (device scalar, may be NULL = n):
 *   number of valid sequences. The input is EITHER x [n,L,64] fp32 (split into hi / lo by the kernel) OR x16
 *   [n,L,P,64] 16-bit planes (hi, lo) as the split-precision tower writes them (then x is ignored).
 *   Packing: svdd_amd/fused.py:pack_gru_lp. */
int svdd_gru_bidir_lp(const float* x, const void* x16, const void* wpack, const float* bpack, const float* inv,
                      float* out, int n, int L, const int32_t* count, int prec, void* stream);

/* svdd_value_tail_lp — svdd_value_tail_f32 with the 64 -> 128 map on the 16-bit matrix cores. w1pack [64 lanes][8 ct][2 c]
 *   [P][8] 16-bit: s_w * W1'[16 ct + j][32 c + 8 g + e]; inv = 1 / s_w; the rest as in the fp32 kernel. `count` as above.
 *   Packing: svdd_amd/fused.py:pack_tail_lp. */
int svdd_value_tail_lp(const float* h_fwd, const float* h_bwd, const void* w1pack, const float* b1,
                       const float* w_eff, const float* b_eff, float inv, float* out, int n, int L, int n_tasks,
                       const int32_t* count, int prec, void* stream);

/* Process-wide options (host). SVDD_OPT_FORCE_EXACT != 0 makes svdd_propose evaluate every draw in the
 * exact (fp64, correctly rounded) arithmetic instead of the filtered fast path — same results, used
 * to A/B the filter. */
enum { SVDD_OPT_FORCE_EXACT = 0,
       SVDD_OPT_MSPLIT = 1 /* tuning: waves per 64-position tile in svdd_propose, 0 = auto */,
       SVDD_OPT_SELECT_ONE_ROW = 2 /* A/B: 1 = svdd_select as one wave per row for every M (default 0: several rows per wave for M <= 64,
                                      4 row groups per wave from 2^21 (row, candidate) slots on); 2 / 3 = force 4 / 1 row groups per wave */,
       SVDD_OPT_BACKBONE_LP_VERSION = 3 /* A/B: 1 = svdd_backbone_cnn_lp runs the round-2 kernel for every shape; 2 (default) = the
                                            transposed-accumulator kernel where one sequence fills a tile (104 < L <= 208);
                                            21 / 22 (= 2) / 23: that kernel with one / two / three waves per SIMD (same bits) */,
       SVDD_OPT_TRUNK_GEMM_VERSION = 4 /* A/B: svdd_trunk_gemm kernel: 1 = 128 x 128 tiles everywhere, 2 (default) = 256 x 256 LDS-DMA
                                           tiles from 128 tiles up, 3 = 256 x 256 everywhere (13 .. 16: timing experiments with
                                           wrong results: no epilogue / one K block / no DMA / no fragment reads); 40 / 41 / 42: the
                                           LDS-DMA kernel's tile height by cost (default) / 256 rows / 192 rows; 51 .. 54: how many
                                           chains of GEMMs share the chip (the cost model prices a launch against CUs / that) */,
       SVDD_OPT_CAND_ROW_STRIDE = 5 /* layout experiment (round 4): bytes between two candidate rows of `cand` as svdd_select /
                                        svdd_select_compact read it (0 = L, the default): rows padded to whole 128-byte lines. The library cannot
                                        check the caller's padding: a non-zero value is refused (SVDD_E_ARG) unless the process has
                                        SVDD_EXPERIMENTS set in its environment */,
       SVDD_OPT_TRUNK_PLANES_F32 = 6 /* the svdd_trunk_* entry points take ONE fp32 operand plane (the *_hi pointers are float*, the
                                         *_lo pointers NULL) and svdd_trunk_gemm multiplies on v_mfma_f32_16x16x4_f32: the Enformer-shaped
                                         trunk at the reference's precision (weights packed by fused_trunk.pack_gemm_weight_f32) */,
       SVDD_OPT_SELECT_BATCHES = 8 /* A/B (round 6): batches of row groups a wave of a SATURATED svdd_select launch (>= 2^21 (row, candidate)
                                      slots, M = 10 / 20) takes: 0 / 1 = one (default), 2 / 4 = the next batch decided under the row
                                      gathers of the one before — measured slower, kept for the record */,
       SVDD_OPT_BACKBONE_SPLIT = 7 /* svdd_backbone_cnn_f32 on several workgroups per sequence (small batches; same bits): 0 = automatic
                                       (4 workgroups per sequence while 4 n <= CUs, 2 while 2 n <= CUs; needs
                                       svdd_backbone_set_workspace), 1 = never, 2 / 4 = that many wherever n R <= CUs */ };
int svdd_set_option(int key, int value);

/* Soak / profiling aid: while `device_counters2` (two zero-initialised uint64 on the device) is non-NULL, every
 * svdd_propose / svdd_sample_categorical launch adds {draws at masked positions, draws the exact-arithmetic filter
 * could not decide on the fast path} to it. NULL switches the counting off. */
int svdd_k1_stats(unsigned long long* device_counters2);

/* Per-launch kernel timing (host). While enabled, svdd_propose (kernel 0), svdd_select (1), svdd_conv1d_cl_f32 (2),
 * svdd_gru_bidir_f32 (3), svdd_epilogue_ln_f32 (4), svdd_conv_tower_f32 (5), svdd_backbone_cnn_f32 / _save_f32 (6), svdd_value_tail_f32 (7), svdd_tds_resample (8), svdd_mt19937_uniform_f32 (9), svdd_backbone_cnn_grad_f32 (10), svdd_gru_bidir_train[2]_f32 (11) and svdd_gru_bidir_bwd[2]_f32 (12; the *2 forms as one span over their two launches) are dispatched (the reward net's DPS kernels share slots: svdd_reward_stem*_f32 5, svdd_reward_tail_grad_f32 7, svdd_conv1d_cl_gated_f32 2)
 * with HIP start/stop events bound to the dispatch on its launch stream
 * (hipExtLaunchKernelGGL); svdd_profile_collect waits for the recorded launches, returns the summed
 * hipEventElapsedTime and their count, and clears the record. Not for use during graph capture. */
int svdd_profile_enable(int on);
int svdd_profile_collect(int kernel, double* total_ms, int* launches);

/* Self-test of the fast-math error bounds the K1 filter relies on (host; synchronous; allocates).
 * out3[0] = max relative error of the fast Gumbel-norm g over ALL 2^24 possible uniforms,
 * out3[1] = max relative error of the fast exp over 2^24 points of [-80,0],
 * out3[2] = max relative error of the fast log over 2^24 points of (1,4]. */
int svdd_selftest_fastmath(double* out3);

/* Library / device probe (host). Returns SVDD_OK and fills arch (e.g. "gfx950") and CU count. */
int svdd_device_info(char* arch, int arch_len, int* num_cu);

/* ABI version of this header: bumped on any signature change. */
int svdd_abi_version(void);
#define SVDD_ABI_VERSION 11

/*
 * Enformer-shaped value trunk (BASELINE.json configs[3]; reference decode.py:78-80, Enformer.py:1271-1334 trunk, :1807-1884
 * conv tower, :1887-2007 transformer tower, :2176-2292 ConvBlock "NACDR") on the 16-bit matrix cores, split precision
 * bf16x3 (a_lo != NULL) or one-pass bf16 (a_lo == NULL). csrc/svdd_trunk.hip. Activations are channels-last rows
 * [n * rows_per_seq, C]; for the k = 5 convolutions rows_per_seq = L + 2 (two zero rows behind every sequence, two guard rows in
 * front of the first), so that a tap is a row shift of a plain GEMM. count (may be NULL): device scalar, number of live sequences of a compacted batch.
 *
 * svdd_trunk_gemm        out[M, N] = act(sum_{t < T} A[rows + t - T/2, Cin] W_t[Cin, N] + bias) (+ resid), fp32 [M, ldo].
 *                        a_hi / a_lo: bf16 operand planes [>= M + 128 + T rows, lda] with T/2 readable rows before row 0;
 *                        w: bf16 weight fragments packed by svdd_amd.fused_trunk.pack_gemm_weight ; N % 128 == 0, Cin % 32 == 0,
 *                        T odd ; act 0 none, 1 relu, 2 x * sigmoid(1.702 x).
 *                        Fused second output (out_hi != NULL; out may then be NULL): the operand planes of the NEXT GEMM,
 *                        post_act(post_scale[c] y + post_shift[c]) -> (out_hi, out_lo) [M, N] (scale / shift NULL: identity), zero in
 *                        the last `pad` rows of every sequence — what svdd_trunk_act_split would write from `out`. The
 *                        output planes must not be the input planes. Two kernels behind it (SVDD_OPT_TRUNK_GEMM_VERSION).
 * svdd_trunk_act_split   x fp32 [rows, C] -> act(scale[c] x + shift[c]) (scale / shift NULL: identity) -> planes hi (lo may be
 *                        NULL) ; the last `pad` rows of every sequence are written as zeros.
 * svdd_trunk_layernorm_split   LayerNorm(x[row, :C]) gamma + beta -> planes (C % 8 == 0, C <= 4096).
 * svdd_trunk_attn_pool   softmax-weighted pooling over position pairs: x, logits fp32 [n, L + 2, C] -> out [n, ceil(L/2) + 2, C]
 *                        (may be NULL) and / or the next GEMM's operand planes post_act(post_scale o + post_shift), pad rows zeroed.
 * svdd_trunk_attn_small  relative-position attention on T <= 4 tokens per sequence (what the conv tower leaves of L <= 512): qkv fp32
 *                        [n T, heads (2 dk + dv)] = [q | k | v], rel_k [heads, 2 T - 1, dk] (positional keys), content / pos bias
 *                        [heads, dk] -> softmax(((q s + cb) k^T + shift((q s + pb) rel_k^T))) v as (hi, lo) planes [n T, heads dv].
 * svdd_trunk_stem_unfold tokens [n, L] u8 -> hi plane [n (L + 2), 64]: channel 4 t + token of position l + t - 7 (t < 15) set to 1.
 *
 * First level shared between a candidate and its parent (exact; the SVDD-MC step of reference diffusion_gosai.py:1203-1209 scores
 * M candidates that differ from x_t at a few positions, and the stem, the 1 x 1 block and the pooling of the first level are
 * functions of a 15-token window): only the rows of up to `slots` (<= 4) even-aligned windows per candidate go through the
 * level's GEMMs, as compact rows without pads; the rest of the level's output planes are copies of the parent's.
 * The next levels go the same way while their lengths are even: level d + 1 can differ in rows [w0/2 - 2, w1/2 + 2) (k = 5
 * convolution); a level d >= 1 works on compact segments of window + 2 rows of context on each side.
 * svdd_trunk_windows         candidate c (cand [n, L] u8, L <= 256) vs row parent_idx[c] / div of parent [., L]: for the `depth`
 *                            shared levels (L % 2^(depth - 1) == 0: only the last may have an odd length) and window slot j: w0[(d n + c) slots + j], wlen[.] = the even-aligned
 *                            windows of level d in ascending order (level 0: one per position that differs, +- halo, windows that
 *                            touch merged, the last slot takes what is left; deeper: windows within 4 rows merged; wlen 0: unused
 *                            slot, and for c >= count), seg[.] = its compact rows (wlen, + 4 context rows for d >= 1).
 * svdd_trunk_stem_unfold_win the stem operand of the window rows: the r-th window row of candidate c at compact row off[c slots] + r
 *                            (off = exclusive prefix sum of seg over [n slots]).
 * svdd_trunk_attn_pool_win   x, logits fp32 (compact rows of a level of length L; the segment of slot s starts at off[s], its window
 *                            in_halo rows further) -> the next GEMM's operand planes: pooled window rows where a window covers the
 *                            pair, else the row of the parent's planes [., L/2 + 2, C] (zeros outside the sequence). v0 == NULL:
 *                            whole sequences [n, L/2 + 2, C], pad rows zero; else the compact segments of the next level: rows
 *                            v0[s] - 2 .. v0[s] + vlen[s] + 1 at off2[s].
 */
int svdd_trunk_gemm(const void* a_hi, const void* a_lo, const void* w, const float* bias, const float* resid, float* out,
                    int M, int N, int Cin, int T, int lda, int ldo, int act, const int32_t* count, int rows_per_seq,
                    void* out_hi, void* out_lo, const float* post_scale, const float* post_shift, int post_act, int pad,
                    void* stream);
int svdd_trunk_act_split(const float* x, const float* scale, const float* shift, int act, int64_t rows, int C,
                         int rows_per_seq, int pad, void* hi, void* lo, const int32_t* count, void* stream);
int svdd_trunk_layernorm_split(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int C,
                               void* hi, void* lo, const int32_t* count, int rows_per_seq, void* stream);
int svdd_trunk_attn_pool(const float* x, const float* logits, int n, int L, int C, float* out, const int32_t* count,
                         void* out_hi, void* out_lo, const float* post_scale, const float* post_shift, int post_act, void* stream);
int svdd_trunk_attn_small(const float* qkv, const float* rel_k, const float* content_bias, const float* pos_bias, int n, int T,
                          int heads, int dk, int dv, void* hi, void* lo, const int32_t* count, void* stream);
int svdd_trunk_stem_unfold(const uint8_t* tok, int n, int L, void* hi, const int32_t* count, void* stream);
int svdd_trunk_windows(const uint8_t* cand, const uint8_t* parent, const int32_t* parent_idx, int div, int n, int L, int halo,
                       int depth, int slots, const int32_t* count, int32_t* w0, int32_t* wlen, int32_t* seg, void* stream);
int svdd_trunk_stem_unfold_win(const uint8_t* tok, int n, int L, int slots, const int32_t* w0, const int32_t* wlen,
                               const int32_t* off, void* hi, const int32_t* count, void* stream);
int svdd_trunk_attn_pool_win(const float* x, const float* logits, int n, int L, int C, int in_halo, int slots, const int32_t* w0,
                             const int32_t* wlen, const int32_t* off, const int32_t* parent_idx, int div, const void* parent_hi,
                             const void* parent_lo, const int32_t* count, void* out_hi, void* out_lo, const float* post_scale,
                             const float* post_shift, int post_act, const int32_t* v0, const int32_t* vlen, const int32_t* off2,
                             void* stream);

/* --------------------------------------------------------------------------------------------------------------------------
 * DPS (gradient guidance, BASELINE configs[4]) without autograd — ABI 11. The reference's step (diffusion_gosai.py:1286-1330):
 *   x_grad = d mean(reward(softmax(E)[:, 0:4])) / d onehot(x_t),  E = keep onehot(x_t) + (1 - keep) log p(x0 | x_t)      (:1321-1330)
 *   q_xs   = exp(log p) (mct - mcs), q_xs[MASK] = mcs, times exp(scale (x_grad - x_grad[MASK]))                        (:1306-1314)
 * Per-position pieces (logits = the one-launch backbone's raw output [B][L][5] contiguous, x = tokens u8 [B][L]):
 *   svdd_dps_probs       -> probs4 [B][L][4] = softmax(E)[..., 0:4], the reward net's input
 *   svdd_dps_probs_bwd   dprobs4 [B][L][4] -> dlogits [B][L][5] (input of svdd_backbone_cnn_grad_f32; zero at unmasked positions) and
 *                        direct [B][L][5] = keep dE (the term through `keep * x_onehot`; zero at masked positions)
 *   svdd_dps_guided_q    grad_backbone + grad_direct = x_grad -> the guided q_xs [B][L][5]
 * The reward net's gradient pass (ConvGRUTrunk + ConvHead, Enformer.py:1411-1426, 2166-2173; eval-mode BatchNorm folded):
 *   svdd_reward_stem_f32 / _bwd_f32   the 4 -> 64 x 15-tap stem on REAL-valued rows x [n][L][4] (w as [15][4][64]): relu(conv + b) /
 *                                     its transpose applied to the gradient at the stem's pre-activation
 *   svdd_conv1d_cl_f32 (act = 1)      a tower layer forwards: relu(conv + b + f_prev)
 *   svdd_conv1d_cl_gated_f32          a tower layer backwards: gate > 0 ? conv^T(g) + f_prev : 0 (wpack = flipped, transposed taps)
 *   svdd_gru_bidir_train_f32 / _bwd_f32   the GRU (above)
 *   svdd_reward_tail_grad_f32         h_fwd, h_bwd [n][L][64] -> d mean_n(mean_l(score)) / d (h_fwd + h_bwd), written to g_fwd AND g_bwd
 *                                     (w1 [128][64], b1 [128]: dense1; gamma / beta [64]: its LayerNorm; w_eff [128]: dense2 and the
 *                                     head collapsed, task 0)
 *   svdd_sum_gate_f32                 g = f > 0 ? a + b : 0 over `count` floats (count % 4 == 0)
 * All: caller-owned device buffers, launched on `stream`, no allocation, no synchronisation. */
int svdd_dps_probs(const float* logits, const uint8_t* x, int B, int L, float* probs4, void* stream);
int svdd_dps_probs_bwd(const float* logits, const uint8_t* x, const float* dprobs4, int B, int L, float* dlogits, float* direct,
                       void* stream);
int svdd_dps_guided_q(const float* logits, const uint8_t* x, const float* grad_backbone, const float* grad_direct, float dm, float mcs,
                      float scale, int B, int L, float* q, void* stream);
int svdd_reward_stem_f32(const float* x, const float* w, const float* b, float* out, int n, int L, int taps, void* stream);
int svdd_reward_stem_bwd_f32(const float* g, const float* w, float* dx, int n, int L, int taps, void* stream);
int svdd_conv1d_cl_gated_f32(const float* x, const float* wpack, float* y, int n, int L, int cin, int cout, int taps, int dilation,
                             const float* f_prev, const float* gate, void* stream);
int svdd_reward_tail_grad_f32(const float* h_fwd, const float* h_bwd, const float* w1, const float* b1, const float* gamma,
                              const float* beta, const float* w_eff, float eps, int n, int L, float* g_fwd, float* g_bwd, void* stream);
int svdd_sum_gate_f32(const float* a, const float* b, const float* f, float* g, int64_t count, void* stream);
/* The GRU pair with the non-recurrent halves off the serial chain (at the DPS batch the chains are latency: 32 workgroups, 200 dependent
 * steps). Same results as svdd_gru_bidir_train_f32 (out / save: same bits) and as dx_fwd + dx_bwd of svdd_gru_bidir_bwd_f32 (to rounding:
 * the two directions accumulate in one chain):
 *   svdd_gru_bidir_train2_f32: gi = caller scratch [2][n L][192] fp32 (receives b + W_i x of every step: one launch on the whole chip),
 *                              then the chain with W_h h only; out [2][n][L][64], save [2][n][L][4][64] as above.
 *   svdd_gru_bidir_bwd2_f32:   da = caller scratch [2][n L][192] (receives the gate derivatives), then
 *                              g [n][L][64] = gate > 0 ? da_fwd W_i,fwd + da_bwd W_i,bwd : 0 (gate [n][L][64] or NULL: no gate). */
int svdd_gru_bidir_train2_f32(const float* x, const float* wpack, const float* bpack, float* gi, float* out, float* save, int n, int L,
                              void* stream);
int svdd_gru_bidir_bwd2_f32(const float* grad_out, const float* out, const float* save, const float* wpack_bwd, float* da, const float* gate,
                            float* g, int n, int L, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SVDD_HIP_H */
