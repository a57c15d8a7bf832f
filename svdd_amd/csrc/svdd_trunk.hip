// svdd_trunk.hip — hand-written kernels for the Enformer-shaped value trunk of BASELINE.json configs[3]
// (reference decode.py:78-80: EnformerTrunk(n_conv=7, channels=1536, n_transformers=11, n_heads=8, key_len=64) +
// ConvHead(1, 3072); layer structure Enformer.py:1271-1334 trunk, :1807-1884 conv tower, :1887-2007 transformer tower,
// :2176-2292 ConvBlock order "NACDR"). Round 2 ran that trunk as an opaque PyTorch / MIOpen module: 60 TFLOP/s.
//
// The trunk is GEMM-shaped from end to end (k = 5 / k = 1 convolutions over 768..1536 channels on 200..2 positions, then
// 11 transformer blocks on 2 tokens), so it goes to the 16-bit matrix cores with the split-precision arithmetic of
// svdd_lp_common.h: every fp32 operand v = hi + lo with hi = bf16(v), lo = bf16(v - hi);
// a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation ("bf16x3", error
// ~5e-7 of sum|a b|), or one pass on hi ("bf16"). bf16 keeps fp32's exponent range: no operand scaling is needed for
// activations of unknown magnitude (that is why the trunk does not offer f16x3).
//
// Data layout: activations are channels-last rows [n * Lp, C], Lp = L + 2: every sequence is followed by two zero rows (the
// two in front of the first sequence are guard rows of the buffer). A k = 5 convolution is then FIVE PLAIN GEMMS ACCUMULATED
// IN ONE KERNEL — K block kb of tap t reads the A tile shifted by t - 2 rows — with no boundary predicate anywhere in the
// GEMM. The pad rows of an fp32 output are garbage; whoever writes the next GEMM's operand planes (a GEMM / pooling epilogue or
// the element-wise pass) writes zeros there. (Round 3a / 3b-1 had two zero rows on EITHER side of a sequence: 12 % of the
// GEMM rows of a forward, half of them at the 4-position level.)
//
//   trunk_gemm_kernel         out[M, N] = act(A[M (+shift), K] W[K, N] + bias) (+ residual), fp32 out
//   trunk_act_split_kernel    fp32 rows -> act(scale * x + shift) -> (hi, lo) planes, pad rows zeroed
//   trunk_ln_split_kernel     LayerNorm over the row -> (hi, lo) planes
//   trunk_attn_pool_kernel    softmax-weighted pooling over position pairs (enformer AttentionPool, pool 2)
//   trunk_stem_unfold_kernel  tokens -> the stem's K = 15 taps x 4 one-hot operand rows (exact in bf16: no lo plane)
// Every kernel takes the number of live sequences as a DEVICE scalar (exact work-skipping: the candidates that are copies
// of their parent never enter the trunk, and no host round trip is needed to know how many are left).
#include "svdd_lp_common.h"

extern "C" int svdd_internal_num_cus();      // svdd_nets.hip

// svdd_set_option(SVDD_OPT_TRUNK_GEMM_VERSION, v): 1 = the 128 x 128 kernel everywhere (A/B), 2 = automatic (default),
// 3 = the 256 x 256 kernel everywhere
static int g_trunk_gemm_version = 2;
static int g_trunk_gemm_big_div = 4;   // automatic: the 256 x 256 kernel from num_cus / div tiles on (A/B: option values 21 .. 36 = div 1 .. 16)
static int g_trunk_gemm_conc = 1;    // option values 51 .. 54: how many chains of GEMMs share the chip (fused_trunk's tower streams): the tile-height
                                     // choice prices a GEMM against CUs / conc
static int g_trunk_gemm_bm = 0;      // A/B (option values 41 / 42 / 40): LDS-DMA tiles 256 rows high everywhere / 192 everywhere / by cost (default)
static int g_trunk_gemm_dbg = 0;     // timing experiments only (13 / 14 / 15 / 16): 256 x 256 kernel without epilogue / with one K block /
                                     // no epilogue + no LDS-DMA in the K loop / no epilogue + no fragment reads in the K loop
extern "C" void svdd_internal_set_trunk_gemm_version(int v) {
  g_trunk_gemm_dbg = (v >= 13 && v <= 16) ? v - 12 : 0;
  g_trunk_gemm_version = (v >= 1 && v <= 3) ? v : (g_trunk_gemm_dbg ? 3 : 2);
  if (v >= 21 && v <= 36) g_trunk_gemm_big_div = v - 20;
  if (v >= 40 && v <= 42) g_trunk_gemm_bm = v == 41 ? 256 : v == 42 ? 192 : 0;
  if (v >= 51 && v <= 54) g_trunk_gemm_conc = v - 50;
}

// svdd_set_option(SVDD_OPT_TRUNK_PLANES_F32, 1): the operand planes are ONE fp32 plane (a_hi / out_hi point at floats, a_lo / out_lo
// must be NULL) and the GEMM multiplies on v_mfma_f32_16x16x4_f32 — the trunk at the reference's own precision (round 4).
static int g_trunk_planes_f32 = 0;
extern "C" void svdd_internal_set_trunk_planes_f32(int v) { g_trunk_planes_f32 = v ? 1 : 0; }

namespace {

typedef __bf16 bf16_t;
typedef b8 BV8;

constexpr int G_BM = 128, G_BN = 128, G_BK = 32;
constexpr int G_AS = 40;                   // halves per LDS row of the A tile (80 B: 64 B of data + 16 B pad)
constexpr int ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2;

__device__ __forceinline__ float gelu_enformer(float x) { return x / (1.0f + __expf(-1.702f * x)); }   // x * sigmoid(1.702 x)
__device__ __forceinline__ float apply_act(float v, int act) {
  return act == ACT_RELU ? fmaxf(v, 0.0f) : act == ACT_GELU ? gelu_enformer(v) : v;
}

struct GemmArgs {
  const bf16_t* a_hi; const bf16_t* a_lo;  // [rows][lda] operand planes; row 0 = logical row 0 (guard rows exist before / after)
  const BV8* w;                            // packed weights [KB = chunk * T + tap][N / 128][8 n-tiles][parts][64 lanes][8]
  const float* bias; const float* resid; float* out;
  int M, N, KB, cb, T, lda, ldo, act;
  const int* count; int rows_per_seq;
  // fused second output: the NEXT GEMM's operand planes p_act(p_scale y + p_shift) -> (hi, lo) [M][N], zero in the `pad`
  // rows at either end of every sequence (what a separate svdd_trunk_act_split pass over y would write)
  bf16_t* o_hi; bf16_t* o_lo; const float* p_scale; const float* p_shift; int p_act, pad;
  int dbg;
  int f32;                                 // planes are fp32 (a_hi / o_hi are float*, no lo planes; lda / N in floats)
};

// ---- operand-plane stores, by plane format: (hi, lo) bf16 halves of an fp32 value, or the fp32 value itself
typedef float f32x8_t __attribute__((ext_vector_type(8)));
typedef bf16_t BV4 __attribute__((ext_vector_type(4)));
template <bool F32>
__device__ __forceinline__ void plane_store8(bf16_t* hi, bf16_t* lo, size_t at, f32x8_t v) {
  if constexpr (F32) {
    float* p = reinterpret_cast<float*>(hi) + at;
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    const BV8 h = __builtin_convertvector(v, BV8);
    *reinterpret_cast<BV8*>(hi + at) = h;
    if (lo) *reinterpret_cast<BV8*>(lo + at) = __builtin_convertvector(v - __builtin_convertvector(h, f32x8_t), BV8);
  }
}
template <bool F32>
__device__ __forceinline__ void plane_store4(bf16_t* hi, bf16_t* lo, size_t at, f32x4 v) {
  if constexpr (F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(hi) + at) = v;
  } else {
    const BV4 h = __builtin_convertvector(v, BV4);
    *reinterpret_cast<BV4*>(hi + at) = h;
    if (lo) *reinterpret_cast<BV4*>(lo + at) = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), BV4);
  }
}
template <bool F32>
__device__ __forceinline__ void plane_store1(bf16_t* hi, bf16_t* lo, size_t at, float o) {
  if constexpr (F32) {
    reinterpret_cast<float*>(hi)[at] = o;
  } else {
    const bf16_t hb = (bf16_t)o;
    hi[at] = hb;
    if (lo) lo[at] = (bf16_t)(o - (float)hb);
  }
}
template <bool F32>
__device__ __forceinline__ void plane_copy4(bf16_t* hi, bf16_t* lo, size_t at, const bf16_t* shi, const bf16_t* slo, size_t sat) {
  if constexpr (F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(hi) + at) = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(shi) + sat);
  } else {
    *reinterpret_cast<BV4*>(hi + at) = *reinterpret_cast<const BV4*>(shi + sat);
    if (lo) *reinterpret_cast<BV4*>(lo + at) = *reinterpret_cast<const BV4*>(slo + sat);
  }
}

// Epilogue of four adjacent columns of one row, shared by both GEMM kernels: y = act(acc + bias) (+ resid) -> fp32 out
// (if any) and / or the operand planes of the next GEMM.
__device__ __forceinline__ void gemm_store4(const GemmArgs& a, int row, int col, bool pad_row, f32x4 v, f32x4 b4, f32x4 ps4, f32x4 pb4) {
  v += b4;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], a.act);
  if (a.resid) v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)row * a.ldo + col);
  if (a.out) *reinterpret_cast<f32x4*>(a.out + (size_t)row * a.ldo + col) = v;
  if (a.o_hi) {
    f32x4 t = v * ps4 + pb4;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = pad_row ? 0.0f : apply_act(a.p_scale ? t[e] : v[e], a.p_act);
    if (a.f32) plane_store4<true>(a.o_hi, nullptr, (size_t)row * a.N + col, t);
    else plane_store4<false>(a.o_hi, a.o_lo, (size_t)row * a.N + col, t);
  }
}
__device__ __forceinline__ bool gemm_pad_row(const GemmArgs& a, int row) {
  if (a.pad <= 0) return false;
  const int pos = row % a.rows_per_seq;
  return pos >= a.rows_per_seq - a.pad;
}

// One workgroup (4 waves) = a 128 x 128 output tile; wave (wm, wn) owns 64 x 64 = 4 x 4 MFMA tiles. A stage = one K block
// (32 wide): the 128 x 32 A tile (shifted by the block's tap) and the 32 x 128 W tile. FOUR stages are in flight: stage s is
// read from one of two LDS buffers by the MFMAs, stage s + 1 moves from registers into the other buffer, the global loads of
// stages s + 2 and s + 3 are outstanding — the A operand streams from HBM (GBs per layer) and one stage of MFMAs does not cover that latency
// (first version: one LDS buffer, loads one stage ahead, two barriers per stage: 350 - 500 TFLOP/s on the MFMA).
// K blocks are ordered channel chunk major, TAP MINOR: the T taps of a chunk read the same 64-byte column of A shifted by one
// row each, T consecutive K blocks, so all but the first come from L1 / L2.
// Workgroup -> tile: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2). The N / 128
// column tiles of one row tile share the A operand, so they are given ids 8 apart: same XCD, back to back.
// The MFMA takes the WEIGHT fragment as its A operand and the activation fragment as B (same registers either way), so a
// lane's accumulator registers are 4 adjacent output columns of one row: 16-byte stores / residual loads in the epilogue.
template <int NPARTS>
__global__ __launch_bounds__(256, 2) void trunk_gemm_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t sA[2][NPARTS][G_BM][G_AS];
  __shared__ __attribute__((aligned(16))) BV8 sB[2][8][NPARTS][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1, j = lane & 15, g = lane >> 4;
  const int m_live = a.count ? min(a.M, *a.count * a.rows_per_seq) : a.M;
  const int NB = a.N / G_BN;
  const int MB = (a.M + G_BM - 1) / G_BM;
  const int lid = blockIdx.x;                               // XCD-aware tile order
  const int grp = lid / (8 * NB), rem = lid - grp * 8 * NB; // 8 row tiles x NB column tiles per group
  const int by = grp * 8 + (rem & 7), nb = rem >> 3;
  if (by >= MB) return;
  const int m0 = by * G_BM;
  if (m0 >= m_live) return;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[i][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // staging registers, two stages: A: thread -> (row tid >> 1, 32-byte half tid & 1); W: 2 NPARTS pieces of 16 B
  const int ar = tid >> 1, ah = tid & 1;
  uint4 ra0[NPARTS][2], ra1[NPARTS][2], ra2[NPARTS][2];      // three register sets: stage k waits in set k % 3
  BV8 rb0[2 * NPARTS], rb1[2 * NPARTS], rb2[2 * NPARTS];
#define TG_LOAD(KB_, RA, RB)                                                                                 \
  { const int kb_ = (KB_);                                                                                   \
    const int c_ = kb_ / a.T, t_ = kb_ - c_ * a.T;                                                           \
    const int64_t row_ = (int64_t)m0 + ar + t_ - a.T / 2;                                                    \
    const size_t off_ = (size_t)(row_ * a.lda + 32 * c_ + 16 * ah);                                          \
    RA[0][0] = *reinterpret_cast<const uint4*>(a.a_hi + off_);                                               \
    RA[0][1] = *reinterpret_cast<const uint4*>(a.a_hi + off_ + 8);                                           \
    if constexpr (NPARTS == 2) {                                                                             \
      RA[1][0] = *reinterpret_cast<const uint4*>(a.a_lo + off_);                                             \
      RA[1][1] = *reinterpret_cast<const uint4*>(a.a_lo + off_ + 8);                                         \
    }                                                                                                        \
    const BV8* src_ = a.w + ((size_t)kb_ * NB + nb) * (8 * NPARTS * 64);                                     \
    _Pragma("unroll") for (int q = 0; q < 2 * NPARTS; ++q) RB[q] = src_[q * 256 + tid]; }
#define TG_STORE(BUF, RA, RB)                                                                                \
  { _Pragma("unroll") for (int p = 0; p < NPARTS; ++p) {                                                     \
      *reinterpret_cast<uint4*>(&sA[BUF][p][ar][16 * ah]) = RA[p][0];                                        \
      *reinterpret_cast<uint4*>(&sA[BUF][p][ar][16 * ah + 8]) = RA[p][1]; }                                  \
    _Pragma("unroll") for (int q = 0; q < 2 * NPARTS; ++q) (&sB[BUF][0][0][0])[q * 256 + tid] = RB[q]; }
  // the three passes of a product go to the same accumulator: a whole row of tiles (4 independent accumulators) between
  // dependent MFMAs
#define TG_COMPUTE(BUF)                                                                                      \
  { BV8 bf[4][NPARTS], af[4][NPARTS];                                                                        \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                         \
      _Pragma("unroll") for (int p = 0; p < NPARTS; ++p) bf[nt][p] = sB[BUF][4 * wn + nt][p][lane];          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                            \
      _Pragma("unroll") for (int p = 0; p < NPARTS; ++p)                                                     \
        af[i][p] = *reinterpret_cast<const BV8*>(&sA[BUF][p][64 * wm + 16 * i + j][8 * g]);                  \
    __builtin_amdgcn_sched_barrier(0);          /* every fragment of the stage is requested before the first MFMA */ \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
      _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                       \
        acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt][0], af[i][0], acc[i][nt], 0, 0, 0);      \
      if constexpr (NPARTS == 2) {                                                                           \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                     \
          acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt][1], af[i][0], acc[i][nt], 0, 0, 0);    \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                     \
          acc[i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt][0], af[i][1], acc[i][nt], 0, 0, 0);    \
      } } }
  const int KB = a.KB;
  TG_LOAD(0, ra0, rb0)
  if (KB > 1) TG_LOAD(1, ra1, rb1)
  if (KB > 2) TG_LOAD(2, ra2, rb2)
  TG_STORE(0, ra0, rb0)
  if (KB > 3) TG_LOAD(3, ra0, rb0)
  __syncthreads();
  // iteration s: MFMAs on stage s (LDS buffer s & 1); stage s + 1 moves from its register set into the other buffer;
  // that set is refilled with stage s + 4: four stages in flight, one barrier per stage
#define TG_ITER(S, BUF, RA, RB)                                                                              \
  if ((S) < KB) {                                                                                            \
    TG_COMPUTE(BUF)                                                                                          \
    if ((S) + 1 < KB) TG_STORE(1 - (BUF), RA, RB)                                                            \
    if ((S) + 4 < KB) TG_LOAD((S) + 4, RA, RB)                                                               \
    __syncthreads();                                                                                         \
  }
  for (int s = 0; s < KB; s += 6) {
    TG_ITER(s, 0, ra1, rb1)
    TG_ITER(s + 1, 1, ra2, rb2)
    TG_ITER(s + 2, 0, ra0, rb0)
    TG_ITER(s + 3, 1, ra1, rb1)
    TG_ITER(s + 4, 0, ra2, rb2)
    TG_ITER(s + 5, 1, ra0, rb0)
  }
#undef TG_ITER
#undef TG_COMPUTE
#undef TG_STORE
#undef TG_LOAD
  // epilogue: lane (j, g) holds row 16 i + j, columns 16 nt + 4 g .. + 3 of every tile
  bool padr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) padr[i] = gemm_pad_row(a, m0 + 64 * wm + 16 * i + j);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int col = nb * G_BN + 64 * wn + 16 * nt + 4 * g;
    f32x4 b4 = {0.0f, 0.0f, 0.0f, 0.0f}, ps4 = {1.0f, 1.0f, 1.0f, 1.0f}, pb4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + col);
    if (a.p_scale) { ps4 = *reinterpret_cast<const f32x4*>(a.p_scale + col); pb4 = *reinterpret_cast<const f32x4*>(a.p_shift + col); }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 64 * wm + 16 * i + j;
      if (row < m_live) gemm_store4(a, row, col, padr[i], acc[i][nt], b4, ps4, pb4);
    }
  }
}

// ------------------------------------------------------------------------------------------ 256 x 256 tiles ----
// Round 3, second GEMM kernel. The 128 x 128 kernel above moves (128 + 128) x 32 x 2 parts x 2 B = 32 KB from L2 into LDS
// for every 1.57 M multiply-adds: at the 2.5 PFLOP/s of the matrix cores that is 26 TB/s, more than the L2s deliver — it ran
// at 370 - 490 TFLOP/s on the MFMA (profiles/r03_trunk_gemms_before.txt), waves parked in s_waitcnt half of the time. Here:
//   * one workgroup (8 waves, one per CU) = a 256 x 256 output tile: twice the multiply-adds per staged byte;
//     wave (wm, wn) owns 128 x 64 = 8 x 4 MFMA tiles (128 accumulator registers);
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass). The DMA
//     writes 1 KB per wave instruction at wave-uniform base + 16 lane, the SOURCE address is per lane: lane 16 g + j
//     fetches bytes 16 g .. + 15 of row j of a 16-row x 32-column sub-tile, so the LDS image of a sub-tile is exactly the
//     MFMA fragment order and every fragment read is ds_read_b128 at base + 16 lane — conflict-free by construction
//     (the 80-byte row pitch of the kernel above spent a third of its LDS cycles on bank conflicts); the weights are
//     stored in that order by the host, their DMA is a straight copy;
//   * LDS holds three stages of the activation operand and two of the weights (5 x 32 KB for hi + lo planes: all 160 KB of
//     a CU): the A tile of K block s + 2 and the W tile of s + 1 are requested while s is multiplied, the wait is counted
//     (the first version had two stages of both and waited vmcnt(0): the DMA was a quarter of the K loop);
//   * the two waves of a SIMD (wm = 0 / 1) run HALF A PHASE APART: a phase = [load segment: fragment reads (+ DMA
//     issue)] barrier [compute segment: 48 MFMAs] barrier, and wave group 1 starts one barrier late, so that on every
//     SIMD one wave multiplies while its partner reads (MI355X_MICROARCH.md, "Two waves per SIMD").
// N need not be a multiple of 256: in a last, half-wide column block the waves with wn >= 2 only stage and synchronise.
constexpr int H_BM = 256, H_BN = 256;

// F32 (round 4; NPARTS = 2): the operand is ONE fp32 plane. A 16-row x 32-column fp32 sub-tile is 2 KB = the two 1 KB pieces
// that the (hi, lo) pair of a bf16 sub-tile occupies, so stages, DMA and barriers are the x3 kernel's; piece p of lane (j, g)
// holds floats 8 g + 4 p .. + 3 of row j (the weights are packed the same way), and the 8 MFMA steps of a 32-wide K block
// multiply k = 8 g + 4 p + e on v_mfma_f32_16x16x4_f32 (the instruction sums over the four lane groups g: every k once).
// 128 MFMAs of 32 cycles per compute segment instead of 48 of 16: the K loop is matrix-pipe bound.
// HT = row sub-tiles of a wave per compute segment: 4 -> a 256-row tile (8 sub-tiles per wave), 3 -> a 192-row tile (6 per wave).
// The 192-row form exists for the GEMMs whose 256-row tiles fill the chip badly (round 4): the transformer tower works on 7680 rows
// at a config-4 step, 30 tiles high — x 6 column tiles = 180 workgroups on 256 CUs (0.70 of a round), x 12 = 360 (1.41 rounds: 2);
// 40 tiles of 192 rows make 240 / 480: 0.94 / 1.88 rounds. The launcher takes whichever height costs fewer round x rows.
template <int NPARTS, bool F32 = false, int HT = 4>
__global__ __launch_bounds__(512, 2) void trunk_gemm256_kernel(GemmArgs a) {
  constexpr int BM = 64 * HT;
  static_assert(!F32 || NPARTS == 2, "fp32 planes: two 16-byte pieces per lane");
  constexpr int ES = F32 ? 4 : 2;                            // bytes per operand element
  extern __shared__ __attribute__((aligned(1024))) char hsm[];
  constexpr int SUB_B = 1024;                               // one 16 x 32 sub-tile of one part
  constexpr int OPER_B = 16 * NPARTS * SUB_B;               // the 16 sub-tiles of an operand: [sub][part][1 KB]
  constexpr int W_BASE = 3 * OPER_B;                        // LDS: three A stages, then two W stages (x3: 5 x 32 KB = all 160 KB)
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3, j = lane & 15, g = lane >> 4;
  const int m_live = a.count ? min(a.M, *a.count * a.rows_per_seq) : a.M;
  const int NB128 = a.N / 128;
  const int NB = (a.N + H_BN - 1) / H_BN;
  const int MB = (a.M + BM - 1) / BM;
  const int lid = blockIdx.x;                               // XCD-aware tile order (as above)
  const int grp = lid / (8 * NB), rem = lid - grp * 8 * NB;
  const int by = grp * 8 + (rem & 7), nb = rem >> 3;
  if (by >= MB) return;
  const int m0 = by * BM;
  if (m0 >= m_live) return;
  const bool cols_ok = 256 * nb + 64 * wn < a.N;            // this wave's 64 columns exist
  const bool wstage_ok = 2 * nb + (w >> 2) < NB128;         // ... and so do the weight sub-tiles it stages
  const bool astage_ok = 2 * w < 4 * HT;                    // the A sub-tiles 2 w, 2 w + 1 exist (HT = 3: waves 0 - 5 stage A)
  f32x4 acc[2 * HT][4];
#pragma unroll
  for (int i = 0; i < 2 * HT; ++i)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[i][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
  // DMA sources: wave w stages the A sub-tiles 2 w, 2 w + 1 (rows m0 + 32 w ..) and the W sub-tiles 2 w, 2 w + 1
  const char* asrc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = min(m0 + 16 * (2 * w + q) + j, a.M - 1);
    asrc[q] = reinterpret_cast<const char*>(a.a_hi) + ((int64_t)row * a.lda + 8 * g) * ES;
  }
  const int64_t lo_off = F32 ? 16 : (NPARTS == 2 ? (a.a_lo - a.a_hi) * 2 : 0);   // bytes from part / piece 0 to 1
  const BV8* wsrc = a.w + ((size_t)(2 * nb + (w >> 2)) * 8 + ((2 * w) & 7)) * (NPARTS * 64) + lane;
  const size_t wstep = (size_t)NB128 * 8 * NPARTS * 64;      // V8s per K block
  char* const stage_a = hsm + (2 * w) * NPARTS * SUB_B;
  char* const stage_w = hsm + W_BASE + (2 * w) * NPARTS * SUB_B;
  // The DMA of a K block is what the K loop waits for (timing experiments 13 / 15, profiles/r03_trunk_gemm_dma.txt: 30.4 ms of
  // K loops with it, 22.8 ms without — 2.16 PFLOP/s): eight 1 KB pieces per wave cost the issuing wave ~150 cycles each
  // inside a load segment, and with two stages in LDS the A tile (an HBM stream) had little more than one segment to land.
  // So: the activation operand gets THREE stages and is requested a whole K block ahead, the weights (L2-resident) two; the
  // four A pieces go out in a K block's second load segment, the four W pieces in its first, and the wait is counted
  // (vmcnt(4): the pieces just issued stay in flight across the barrier).
#define H_DMA_A(KB_, ABUF)                                                                                   \
  if (astage_ok) { const int kb_ = (KB_);                                                                    \
    const int c_ = kb_ / a.T, t_ = kb_ - c_ * a.T;                                                           \
    const int64_t koff_ = ((int64_t)(t_ - a.T / 2) * a.lda + 32 * c_) * ES;                                  \
    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                            \
      _Pragma("unroll") for (int p = 0; p < NPARTS; ++p)                                                     \
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(asrc[q] + koff_ + p * lo_off),                          \
                                         (lds_ptr_t)(stage_a + (ABUF) * OPER_B + (q * NPARTS + p) * SUB_B), 16, 0, 0); }
#define H_DMA_W(KB_, WBUF)                                                                                   \
  if (wstage_ok) {                                                                                           \
    const BV8* ws_ = wsrc + (size_t)(KB_) * wstep;                                                           \
    _Pragma("unroll") for (int q = 0; q < 2 * NPARTS; ++q)                                                   \
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ws_ + q * 64),                                            \
                                       (lds_ptr_t)(stage_w + (WBUF) * OPER_B + q * SUB_B), 16, 0, 0);        \
  }
  const char* const frag_a = hsm + (2 * HT * wm) * NPARTS * SUB_B + 16 * lane;
  const char* const frag_w = hsm + W_BASE + (4 * wn) * NPARTS * SUB_B + 16 * lane;
  f32x4 bf[4][NPARTS], af[4][NPARTS];                       // 16-byte fragments: 8 bf16, or 4 floats (F32)
#define H_READ_W(WBUF)                                                                                       \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                           \
    _Pragma("unroll") for (int p = 0; p < NPARTS; ++p)                                                       \
      bf[nt][p] = *reinterpret_cast<const f32x4*>(frag_w + (WBUF) * OPER_B + (nt * NPARTS + p) * SUB_B);
#define H_READ_A(ABUF, H)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < HT; ++i)                                                             \
    _Pragma("unroll") for (int p = 0; p < NPARTS; ++p)                                                       \
      af[i][p] = *reinterpret_cast<const f32x4*>(frag_a + (ABUF) * OPER_B + ((HT * (H) + i) * NPARTS + p) * SUB_B);
  // the three passes of a product go to the same accumulator: 16 independent MFMAs between dependent ones
#define H_BF(V) __builtin_bit_cast(BV8, V)
#define H_MFMA(H)                                                                                            \
  if (cols_ok) {                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    if constexpr (F32) {                                                                                     \
      _Pragma("unroll") for (int p = 0; p < 2; ++p)                                                          \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                        \
          _Pragma("unroll") for (int i = 0; i < HT; ++i)                                                      \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                 \
              acc[HT * (H) + i][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[nt][p][e], af[i][p][e], acc[HT * (H) + i][nt], 0, 0, 0); \
    } else {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < HT; ++i)                                                            \
      _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                       \
        acc[HT * (H) + i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H_BF(bf[nt][0]), H_BF(af[i][0]), acc[HT * (H) + i][nt], 0, 0, 0); \
    if constexpr (NPARTS == 2) {                                                                             \
      _Pragma("unroll") for (int i = 0; i < HT; ++i)                                                          \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                     \
          acc[HT * (H) + i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H_BF(bf[nt][1]), H_BF(af[i][0]), acc[HT * (H) + i][nt], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < HT; ++i)                                                          \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                     \
          acc[HT * (H) + i][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H_BF(bf[nt][0]), H_BF(af[i][1]), acc[HT * (H) + i][nt], 0, 0, 0); \
    }                                                                                                        \
    }                                                                                                        \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  }
#define H_BARRIER() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
  // K block S (A stage ABUF = S % 3, W stage WBUF = S % 2): two phases of [load segment | barrier | compute segment | barrier]
#define H_KBLOCK(S, ABUF, WBUF)                                                                              \
  if ((S) < KB) {                                                                                            \
    if (cols_ok && (a.dbg != 4 || (S) == 0)) { H_READ_W(WBUF) H_READ_A(ABUF, 0) }                            \
    if ((S) + 1 < KB && a.dbg != 3) H_DMA_W((S) + 1, 1 - (WBUF))                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
    H_BARRIER()                                                                                              \
    H_MFMA(0)                                                                                                \
    H_BARRIER()                                                                                              \
    if (cols_ok && a.dbg != 4) { H_READ_A(ABUF, 1) }                                                         \
    if ((S) + 2 < KB && a.dbg != 3) {                                                                        \
      H_DMA_A((S) + 2, ((ABUF) + 2) % 3)                                                                     \
      if (astage_ok) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(2 * NPARTS) : "memory");   /* K block S + 1 has landed; S + 2's A stays in flight */ \
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                /* (a wave without A pieces: only its W pieces are out) */ \
    } else {                                                                                                 \
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                            \
    }                                                                                                        \
    H_BARRIER()                                                                                              \
    H_MFMA(1)                                                                                                \
    H_BARRIER()                                                                                              \
  }
  const int KB = a.dbg == 2 ? 1 : a.KB;
  H_DMA_A(0, 0)
  H_DMA_W(0, 0)
  if (KB > 1) {
    H_DMA_A(1, 1)
    if (astage_ok) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NPARTS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  H_BARRIER()
  if (wm == 1) H_BARRIER()                                  // wave group 1 runs one segment behind group 0
  for (int s = 0; s < KB; s += 6) {
    H_KBLOCK(s, 0, 0)
    H_KBLOCK(s + 1, 1, 1)
    H_KBLOCK(s + 2, 2, 0)
    H_KBLOCK(s + 3, 0, 1)
    H_KBLOCK(s + 4, 1, 0)
    H_KBLOCK(s + 5, 2, 1)
  }
  if (wm == 0) H_BARRIER()
#undef H_KBLOCK
#undef H_BARRIER
#undef H_MFMA
#undef H_BF
#undef H_READ_A
#undef H_READ_W
#undef H_DMA_W
#undef H_DMA_A
  // epilogue. Lane (j, g) holds row 16 i + j, columns 16 nt + 4 g .. + 3 of every tile of its wave: stored from there, a
  // wave instruction touches 16 rows x 64 B (fp32) or 16 rows x 32 B (planes) — partial cache lines, and the epilogues ran at
  // 1.9 - 3.5 TB/s (profiles/r03_trunk_gemm_epilogue.txt: 17.7 ms of the 45 ms of GEMMs). The accumulators are therefore
  // turned through a wave-private 64-row x 64-column LDS slab (the K-block buffers are dead by now; row pitch 68 floats:
  // conflict-free 16-byte writes and reads) and leave row-contiguously: 4 rows x 256 B per fp32 store, 4 rows x 128 B per
  // plane store, residual rows read the same way. Same arithmetic per element as gemm_store4 from the accumulator layout.
  if (!cols_ok || a.dbg == 1 || a.dbg >= 3) return;
  constexpr int SP = 68;
  float* const slab = reinterpret_cast<float*>(hsm) + w * (64 * SP);
  const int rsub = lane >> 4, c4 = (lane & 15) * 4;
  const int col = nb * H_BN + 64 * wn + c4;
  f32x4 b4 = {0.0f, 0.0f, 0.0f, 0.0f}, ps4 = {1.0f, 1.0f, 1.0f, 1.0f}, pb4 = {0.0f, 0.0f, 0.0f, 0.0f};
  if (a.bias) b4 = *reinterpret_cast<const f32x4*>(a.bias + col);
  if (a.p_scale) { ps4 = *reinterpret_cast<const f32x4*>(a.p_scale + col); pb4 = *reinterpret_cast<const f32x4*>(a.p_shift + col); }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < HT; ++i)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        *reinterpret_cast<f32x4*>(slab + (16 * i + j) * SP + 16 * nt + 4 * g) = acc[HT * h + i][nt];
    const int rbase = m0 + 32 * HT * wm + 16 * HT * h + rsub;
    int pos = a.pad > 0 ? rbase % a.rows_per_seq : 0;
#pragma unroll
    for (int p = 0; p < 4 * HT; ++p) {
      const int row = rbase + 4 * p;
      const f32x4 v = *reinterpret_cast<const f32x4*>(slab + (4 * p + rsub) * SP + c4);
      const bool pad_row = a.pad > 0 && pos >= a.rows_per_seq - a.pad;
      if (row < m_live) gemm_store4(a, row, col, pad_row, v, b4, ps4, pb4);
      pos += 4;
      if (pos >= a.rows_per_seq) pos -= a.rows_per_seq;
    }
  }
}

// fp32 rows [rows, C] -> act(scale[c] x + shift[c]) -> (hi, lo) bf16 planes [rows, C]; rows whose position inside their
// sequence (row % rows_per_seq) lies in its last `pad` rows become zero. One thread = 8 adjacent channels.
struct ActArgs {
  const float* x; const float* scale; const float* shift; int act; int64_t rows; int C, rows_per_seq, pad;
  bf16_t* hi; bf16_t* lo; const int* count;
};
template <bool F32>
__global__ __launch_bounds__(256) void trunk_act_split_kernel(ActArgs a) {
  const int64_t live = a.count ? min(a.rows, (int64_t)*a.count * a.rows_per_seq) : a.rows;
  const int c8 = a.C >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= live * c8) return;
  const int64_t row = idx / c8;
  const int c = (int)(idx - row * c8) * 8;
  const int pos = (int)(row % a.rows_per_seq);
  f32x8_t v;
  if (pos >= a.rows_per_seq - a.pad) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
  } else {
    const f32x4* px = reinterpret_cast<const f32x4*>(a.x + row * a.C + c);
    const f32x4 x0 = px[0], x1 = px[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = e < 4 ? x0[e] : x1[e - 4];
      if (a.scale) t = t * a.scale[c + e] + a.shift[c + e];
      v[e] = apply_act(t, a.act);
    }
  }
  plane_store8<F32>(a.hi, a.lo, (size_t)(row * a.C + c), v);
}

// LayerNorm over a row of C <= 4096 channels (two-pass mean / centred variance, like ATen) -> (hi, lo) planes.
// One wave per row; lane l holds channels 8 (l + 64 q) .. + 7.
struct LnArgs { const float* x; const float* gamma; const float* beta; float eps; int64_t rows; int C; bf16_t* hi; bf16_t* lo; const int* count; int rows_per_seq; };
template <bool F32>
__global__ __launch_bounds__(256) void trunk_ln_split_kernel(LnArgs a) {
  const int64_t live = a.count ? min(a.rows, (int64_t)*a.count * a.rows_per_seq) : a.rows;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= live) return;
  const int lane = threadIdx.x & 63;
  constexpr int QMAX = 8;                                   // 64 lanes x 8 channels x 8 = 4096
  const int nq = (a.C + 511) / 512;
  f32x8_t v[QMAX];
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < QMAX; ++q) {
    const int c = 8 * (lane + 64 * q);
    if (q < nq && c < a.C) {
      const f32x4* px = reinterpret_cast<const f32x4*>(a.x + row * a.C + c);
      const f32x4 x0 = px[0], x1 = px[1];
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[q][e] = e < 4 ? x0[e] : x1[e - 4]; s += v[q][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[q][e] = 0.0f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  const float mean = s / (float)a.C;
  float q2 = 0.0f;
#pragma unroll
  for (int q = 0; q < QMAX; ++q) {
    const int c = 8 * (lane + 64 * q);
    if (q < nq && c < a.C) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[q][e] - mean; q2 += d * d; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q2 += __shfl_xor(q2, off, 64);
  const float rs = rsqrtf(q2 / (float)a.C + a.eps);
#pragma unroll
  for (int q = 0; q < QMAX; ++q) {
    const int c = 8 * (lane + 64 * q);
    if (q < nq && c < a.C) {
      f32x8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[q][e] - mean) * rs * a.gamma[c + e] + a.beta[c + e];
      plane_store8<F32>(a.hi, a.lo, (size_t)(row * a.C + c), o);
    }
  }
}

// AttentionPool(pool_size = 2): out[b, i, c] = sum_k x[b, 2 i + k, c] softmax_k(logits[b, 2 i + k, c]); an odd L is padded
// with one masked position (weight 0). x / logits [n, L + 2, C] and out [n, ceil(L / 2) + 2, C] in the padded layout.
// Outputs (either may be NULL): the fp32 rows (their pad rows are left untouched) and / or the operand planes of the next
// GEMM, act(scale o + shift) -> (hi, lo), INCLUDING zeroed pad rows (what svdd_trunk_act_split would write from the fp32 rows).
struct PoolArgs { const float* x; const float* logits; int n, L, C; float* out; const int* count;
                  bf16_t* hi; bf16_t* lo; const float* scale; const float* shift; int act; };
template <bool F32>
__global__ __launch_bounds__(256) void trunk_attn_pool_kernel(PoolArgs a) {
  const int nlive = a.count ? min(a.n, *a.count) : a.n;
  const int Lo = (a.L + 1) / 2, c4 = a.C >> 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)nlive * (Lo + 2) * c4) return;
  const int c = (int)(idx % c4) * 4;
  const int64_t t = idx / c4;
  const int i = (int)(t % (Lo + 2));                         // output position; Lo, Lo + 1 are the pad rows
  const int64_t b = t / (Lo + 2);
  const int64_t orow = b * (Lo + 2) + i;
  f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool real = i < Lo;
  if (real) {
    const int64_t r0 = b * (a.L + 2) + 2 * i;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(a.x + r0 * a.C + c), l0 = *reinterpret_cast<const f32x4*>(a.logits + r0 * a.C + c);
    o = x0;
    if (2 * i + 1 < a.L) {
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.x + (r0 + 1) * a.C + c), l1 = *reinterpret_cast<const f32x4*>(a.logits + (r0 + 1) * a.C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m = fmaxf(l0[e], l1[e]);
        const float e0 = __expf(l0[e] - m), e1 = __expf(l1[e] - m);
        o[e] = (x0[e] * e0 + x1[e] * e1) / (e0 + e1);
      }
    }
    if (a.out) *reinterpret_cast<f32x4*>(a.out + orow * a.C + c) = o;
  }
  if (a.hi) {
    f32x4 tt = {0.0f, 0.0f, 0.0f, 0.0f};
    if (real) {
#pragma unroll
      for (int e = 0; e < 4; ++e) tt[e] = apply_act(a.scale ? o[e] * a.scale[c + e] + a.shift[c + e] : o[e], a.act);
    }
    plane_store4<F32>(a.hi, a.lo, (size_t)(orow * a.C + c), tt);
  }
}

// Relative-position attention of the transformer tower on the T <= 4 tokens a sequence is pooled down to (2 for L = 200;
// enformer-pytorch Attention as restated in svdd_amd/enformer_value.py RelPosAttention): one wave per (sequence, head),
//   logits[i][j] = (q_i s + content_bias) . k_j + (q_i s + pos_bias) . rel_k[j - i + T - 1],  s = dk^-1/2
//   out_i = sum_j softmax_j(logits[i][.]) v_j  -> written as the (hi, lo) operand planes of the output projection.
// (Round 2/3a ran these as five batched torch matmul / einsum calls per block: two skinny rocBLAS kernels, 0.76 ms per
// block at 3840 sequences for 0.1 GFLOP — profiles/r03_trunk_kernel_split.txt.)
struct AttnArgs {
  const float* qkv; const float* rel_k; const float* content_bias; const float* pos_bias;
  int n, h, dk, dv, ld; float scale; bf16_t* hi; bf16_t* lo; const int* count;
};
template <int T, bool F32>
__global__ __launch_bounds__(256) void trunk_attn_small_kernel(AttnArgs a) {
  const int nlive = a.count ? min(a.n, *a.count) : a.n;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = wid / a.h, hd = wid - b * a.h;
  if (b >= nlive) return;
  const int nq = a.h * a.dk;
  const float* row0 = a.qkv + (size_t)b * T * a.ld;
  float lg[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int jj = 0; jj < T; ++jj) lg[i][jj] = 0.0f;
  for (int d = lane; d < a.dk; d += 64) {
    const float cb = a.content_bias[hd * a.dk + d], pb = a.pos_bias[hd * a.dk + d];
    float kk[T], rk[2 * T - 1];
#pragma unroll
    for (int jj = 0; jj < T; ++jj) kk[jj] = row0[(size_t)jj * a.ld + nq + hd * a.dk + d];
#pragma unroll
    for (int r = 0; r < 2 * T - 1; ++r) rk[r] = a.rel_k[((size_t)hd * (2 * T - 1) + r) * a.dk + d];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      const float q = row0[(size_t)i * a.ld + hd * a.dk + d] * a.scale;
#pragma unroll
      for (int jj = 0; jj < T; ++jj) lg[i][jj] += (q + cb) * kk[jj] + (q + pb) * rk[jj - i + T - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < T; ++i)
#pragma unroll
    for (int jj = 0; jj < T; ++jj)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) lg[i][jj] += __shfl_xor(lg[i][jj], off, 64);
  float pr[T][T];
#pragma unroll
  for (int i = 0; i < T; ++i) {
    float m = lg[i][0];
#pragma unroll
    for (int jj = 1; jj < T; ++jj) m = fmaxf(m, lg[i][jj]);
    float sum = 0.0f;
#pragma unroll
    for (int jj = 0; jj < T; ++jj) { pr[i][jj] = __expf(lg[i][jj] - m); sum += pr[i][jj]; }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jj = 0; jj < T; ++jj) pr[i][jj] *= inv;
  }
  const int ldo = a.h * a.dv;
  for (int c = lane; c < a.dv; c += 64) {
    float vv[T];
#pragma unroll
    for (int jj = 0; jj < T; ++jj) vv[jj] = row0[(size_t)jj * a.ld + 2 * nq + hd * a.dv + c];
#pragma unroll
    for (int i = 0; i < T; ++i) {
      float o = 0.0f;
#pragma unroll
      for (int jj = 0; jj < T; ++jj) o += pr[i][jj] * vv[jj];
      const size_t at = ((size_t)b * T + i) * ldo + hd * a.dv + c;
      plane_store1<F32>(a.hi, a.lo, at, o);
    }
  }
}

// Stem operand: row (b, l) of the padded layout gets the 64 channels [tap t = 0..14][one-hot 4] (+ 4 zeros) of the k = 15
// convolution: channel 4 t + tok[l + t - 7]. Exact in bf16, so the stem GEMM needs A_hi x (W_hi + W_lo) only.
struct StemArgs { const uint8_t* tok; int n, L; bf16_t* hi; const int* count; };
template <bool F32>
__global__ __launch_bounds__(256) void trunk_stem_unfold_kernel(StemArgs a) {
  const int nlive = a.count ? min(a.n, *a.count) : a.n;
  const int Lp = a.L + 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one thread = 8 channels (two taps) of one row
  if (idx >= (int64_t)nlive * Lp * 8) return;
  const int q = (int)(idx & 7);
  const int64_t row = idx >> 3;
  const int pos = (int)(row % Lp);
  const int64_t b = row / Lp;
  f32x8_t v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.0f;
  if (pos >= 0 && pos < a.L) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = 2 * q + h, p = pos + t - 7;
      if (t < 15 && p >= 0 && p < a.L) {
        const int tk = a.tok[b * a.L + p];
        if (tk < 4) v[4 * h + tk] = 1.0f;
      }
    }
  }
  plane_store8<F32>(a.hi, nullptr, (size_t)(row * 64 + 8 * q), v);
}

// ---- first level shared between a candidate and its parent (exact) ------------------------------------------------------
// The stem (k = 15), the 1 x 1 residual block and the pooling logits of the first level make row p of a sequence a function
// of tokens p - 7 .. p + 7 only, and a pooled row i a function of rows 2 i, 2 i + 1. A candidate of the SVDD-MC step differs
// from its parent x_t at a few positions: only the rows of ONE even-aligned window [w0, w0 + wlen) around them are computed
// (compact rows, no pads: every GEMM of the level is 1 x 1 over the unfolded stem operand), the rest of the level's output
// planes are the parent's, copied. Same kernels and the same per-row arithmetic as the whole-sequence path: same bits.
// The next levels go the same way (depth levels in all, each of even length): the rows of level d + 1 that can differ are
// [w0 / 2 - 2, w1 / 2 + 2) (k = 5 convolution), again even-aligned; a level d >= 1 works on compact SEGMENTS of window + 2 rows
// of context on each side (what the k = 5 taps read: pooled window rows, the parent's rows, zeros outside the sequence —
// the outputs at the context rows are garbage and never read). The last shared level writes whole-sequence planes (and may
// have an odd length: its last row is then pooled alone, as in trunk_attn_pool_kernel).
//   trunk_windows_kernel        one wave per live candidate: first / last position that differs from the parent -> w0, wlen
//   trunk_stem_unfold_win_kernel  the stem operand of the window rows, at compact row off[c] + r
//   trunk_attn_pool_win_kernel  pooling of the window rows + copy of the parent's planes elsewhere -> the next level's operands
constexpr int WIN_K = 4;                                     // window slots per candidate and level
struct WinArgs { const uint8_t* cand; const uint8_t* parent; const int* pidx; int div, n, L, halo, depth, K; const int* count;
                 int* w0; int* wlen; int* seg; };
__global__ __launch_bounds__(256) void trunk_windows_kernel(WinArgs a) {
  const int nlive = a.count ? min(a.n, *a.count) : a.n;
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= a.n) return;
  // the positions that differ, as bit masks in ascending order: bit `lane` of mask r = position 64 r + lane
  unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};
  if (c < nlive) {
    const uint8_t* cr = a.cand + (size_t)c * a.L;
    const uint8_t* pr = a.parent + (size_t)(a.pidx[c] / a.div) * a.L;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = 64 * r + lane;
      m[r] = __ballot(p < a.L && cr[p] != pr[p]);
    }
  }
  if (lane != 0) return;
  const int K = a.K;
  int s0[WIN_K], s1[WIN_K], ns = 0;
  // level 0: one window [p - halo, p + halo] per changed position, even-aligned; windows that touch are merged, and from the
  // K-th window on everything goes into the last slot
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned long long mm = m[r];
    while (mm) {
      const int p = 64 * r + __builtin_ctzll(mm);
      mm &= mm - 1;
      const int lo = max(0, p - a.halo) & ~1, hi = min(a.L, (p + a.halo + 2) & ~1);
      if (ns && (lo <= s1[ns - 1] || ns == K)) s1[ns - 1] = max(s1[ns - 1], hi);
      else { s0[ns] = lo; s1[ns] = hi; ++ns; }
    }
  }
  int Lc = a.L;
  for (int d = 0; d < a.depth; ++d) {
    for (int j = 0; j < K; ++j) {
      const int wl = j < ns ? s1[j] - s0[j] : 0;
      const size_t at = ((size_t)d * a.n + c) * K + j;
      a.w0[at] = j < ns ? s0[j] : 0; a.wlen[at] = wl;
      a.seg[at] = wl > 0 ? wl + (d ? 4 : 0) : 0;             // compact rows: the window (+ 2 rows of context on each side)
    }
    // level d + 1: rows / 2, two more on each side (k = 5); a window within 4 rows of the one before joins it (two
    // segments would cost 4 context rows more)
    Lc >>= 1;
    int nn = 0;
    for (int j = 0; j < ns; ++j) {
      const int lo = max(0, (s0[j] >> 1) - 2) & ~1, hi = min(Lc, ((s1[j] >> 1) + 3) & ~1);
      if (nn && lo <= s1[nn - 1] + 4) s1[nn - 1] = max(s1[nn - 1], hi);
      else { s0[nn] = lo; s1[nn] = hi; ++nn; }
    }
    ns = nn;
  }
}

struct StemWinArgs { const uint8_t* tok; int n, L, K; const int* w0; const int* wlen; const int* off; bf16_t* hi; const int* count; };
template <bool F32>
__global__ __launch_bounds__(256) void trunk_stem_unfold_win_kernel(StemWinArgs a) {
  const int nlive = a.count ? min(a.n, *a.count) : a.n;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;       // one thread = 8 channels (two taps) of one row
  const int q = (int)(idx & 7);
  const int64_t t = idx >> 3;
  int r = (int)(t % a.L);                                            // r-th window row of the candidate (its windows are disjoint)
  const int64_t b = t / a.L;
  if (b >= nlive) return;
  int pos = -1;
  const int64_t row = (int64_t)a.off[b * a.K] + r;                   // level 0 has no context rows: the slots' rows are consecutive
  for (int j = 0; j < a.K; ++j) {
    const int wl = a.wlen[b * a.K + j];
    if (r < wl) { pos = a.w0[b * a.K + j] + r; break; }
    r -= wl;
  }
  if (pos < 0) return;
  f32x8_t v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.0f;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int tp = 2 * q + h, p = pos + tp - 7;
    if (tp < 15 && p >= 0 && p < a.L) {
      const int tk = a.tok[b * a.L + p];
      if (tk < 4) v[4 * h + tk] = 1.0f;
    }
  }
  plane_store8<F32>(a.hi, nullptr, (size_t)(row * 64 + 8 * q), v);
}

struct PoolWinArgs { const float* x; const float* logits; int n, L, C, in_halo, K; const int* w0; const int* wlen; const int* off;
                     const int* pidx; int div; const bf16_t* p_hi; const bf16_t* p_lo; const int* count;
                     bf16_t* hi; bf16_t* lo; const float* scale; const float* shift; int act;
                     const int* v0; const int* vlen; const int* off2; };
template <bool F32>
__global__ __launch_bounds__(256) void trunk_attn_pool_win_kernel(PoolWinArgs a) {
  // one workgroup per live candidate: it walks the candidate's output rows (a grid over the upper bound of rows per candidate
  // was 85 M threads for 0.4 M rows at the first level: 0.33 ms per launch)
  const int nlive = a.count ? min(a.n, *a.count) : a.n;
  const int64_t b = blockIdx.x;
  if (b >= nlive) return;
  const int Lo = (a.L + 1) / 2, c4 = a.C >> 2;               // an odd L (last shared level only) pairs its last row with a masked one
  int nrows = Lo + 2;
  if (a.v0) {
    nrows = 0;
    for (int j = 0; j < a.K; ++j) { const int vl = a.vlen[b * a.K + j]; nrows += vl > 0 ? vl + 4 : 0; }
  }
  for (int e = threadIdx.x; e < nrows * c4; e += 256) {
  const int c = (e % c4) * 4;
  int r = e / c4;
  int i = -1000;                                             // output position: Lo, Lo + 1 are the pad rows of the whole-sequence layout
  int64_t orow = 0;
  if (a.v0) {                                                // compact output: rows v0 - 2 .. v0 + vlen + 1 of every window of the next level
    for (int j = 0; j < a.K; ++j) {
      const int vl = a.vlen[b * a.K + j];
      const int len = vl > 0 ? vl + 4 : 0;
      if (r < len) { i = a.v0[b * a.K + j] - 2 + r; orow = (int64_t)a.off2[b * a.K + j] + r; break; }
      r -= len;
    }
  } else {
    i = r;
    orow = b * (Lo + 2) + r;
  }
  int64_t r0 = -1;                                           // the compact row of this level that holds row 2 i, if a window covers it
  if (i >= 0 && i < Lo)
    for (int j = 0; j < a.K; ++j) {
      const int w0 = a.w0[b * a.K + j], wl = a.wlen[b * a.K + j];
      if (2 * i >= w0 && 2 * i < w0 + wl) { r0 = (int64_t)a.off[b * a.K + j] + a.in_halo + 2 * i - w0; break; }
    }
  if (r0 >= 0) {
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(a.x + r0 * a.C + c), l0 = *reinterpret_cast<const f32x4*>(a.logits + r0 * a.C + c);
    f32x4 o = x0;
    if (2 * i + 1 < a.L) {
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(a.x + (r0 + 1) * a.C + c), l1 = *reinterpret_cast<const f32x4*>(a.logits + (r0 + 1) * a.C + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float m = fmaxf(l0[e], l1[e]);
        const float e0 = __expf(l0[e] - m), e1 = __expf(l1[e] - m);
        o[e] = (x0[e] * e0 + x1[e] * e1) / (e0 + e1);
      }
    }
    f32x4 tt;
#pragma unroll
    for (int e = 0; e < 4; ++e) tt[e] = apply_act(a.scale ? o[e] * a.scale[c + e] + a.shift[c + e] : o[e], a.act);
    plane_store4<F32>(a.hi, a.lo, (size_t)(orow * a.C + c), tt);
  } else {                                                   // the parent's row (its pad rows are zero: also what lies outside the sequence)
    const int ip = i < 0 ? Lo : min(i, Lo);
    const int64_t prow = (int64_t)(a.pidx[b] / a.div) * (Lo + 2) + ip;
    plane_copy4<F32>(a.hi, a.lo, (size_t)(orow * a.C + c), a.p_hi, a.p_lo, (size_t)(prow * a.C + c));
  }
  }
}

}  // namespace

extern "C" {

int svdd_trunk_gemm(const void* a_hi, const void* a_lo, const void* w, const float* bias, const float* resid, float* out,
                    int M, int N, int Cin, int T, int lda, int ldo, int act, const int32_t* count, int rows_per_seq,
                    void* out_hi, void* out_lo, const float* post_scale, const float* post_shift, int post_act, int pad,
                    void* stream) {
  if (!a_hi || !w || (!out && !out_hi) || M <= 0 || N <= 0 || (N % G_BN) || Cin <= 0 || (Cin % G_BK) || T < 1 || !(T & 1) ||
      lda < Cin || ldo < N || act < 0 || act > 2 || ((count || pad > 0) && rows_per_seq <= 0) || (out_lo && !out_hi) ||
      ((post_scale == nullptr) != (post_shift == nullptr)) || post_act < 0 || post_act > 2 || pad < 0 ||
      out_hi == a_hi || (out_lo && out_lo == a_lo))
    return SVDD_E_ARG;
  GemmArgs a{(const bf16_t*)a_hi, (const bf16_t*)a_lo, (const BV8*)w, bias, resid, out, M, N, T * (Cin / G_BK), Cin / G_BK, T,
             lda, ldo, act, count, rows_per_seq, (bf16_t*)out_hi, (bf16_t*)out_lo, post_scale, post_shift, post_act, pad, g_trunk_gemm_dbg,
             g_trunk_planes_f32};
  // tile height of the LDS-DMA kernel: 256 rows, or 192 where that costs fewer (rounds of the chip) x (rows per tile)
  const int ncu = svdd_internal_num_cus();
  const int nb2 = (N + H_BN - 1) / H_BN;
  const int ncu_eff = ncu / g_trunk_gemm_conc > 0 ? ncu / g_trunk_gemm_conc : 1;
  auto cost = [&](int bm) { const int64_t tiles = (int64_t)((M + bm - 1) / bm) * nb2; return ((tiles + ncu_eff - 1) / ncu_eff) * bm; };
  const bool short_tiles = g_trunk_gemm_bm == 192 || (g_trunk_gemm_bm == 0 && cost(192) < cost(256));
  const int bm = short_tiles ? 192 : 256;
  const int mb2 = (M + bm - 1) / bm;
  const dim3 grid2((unsigned)(((mb2 + 7) / 8) * 8 * nb2));
#define SVDD_GEMM256(NP_, F32_, LDS_)                                                                                        \
  { if (short_tiles) {                                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(trunk_gemm256_kernel<NP_, F32_, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_); \
      hipLaunchKernelGGL((trunk_gemm256_kernel<NP_, F32_, 3>), grid2, dim3(512), LDS_, (hipStream_t)stream, a);              \
    } else {                                                                                                                \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(trunk_gemm256_kernel<NP_, F32_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_); \
      hipLaunchKernelGGL((trunk_gemm256_kernel<NP_, F32_, 4>), grid2, dim3(512), LDS_, (hipStream_t)stream, a);              \
    } }
  if (g_trunk_planes_f32) {
    // fp32 planes: one plane, fp32 MFMA, always the LDS-DMA kernel (its K loop is matrix-pipe bound at any tile count)
    if (a_lo || out_lo) return SVDD_E_ARG;
    SVDD_GEMM256(2, true, 5 * 16 * 2 * 1024)
    return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
  }
  // 256 x 256 tiles from half a chip's worth of tiles on (the 7680-row GEMMs of the transformer tower make 180 - 360 of them and
  // still run 2.3x faster than on 128 x 128 tiles: 0.79 - 0.98 vs 0.37 - 0.42 PFLOP/s); the 128 x 128 kernel below that
  const bool big = g_trunk_gemm_version == 2 ? ((int64_t)((M + 255) / 256) * nb2 >= ncu / g_trunk_gemm_big_div)
                                              : g_trunk_gemm_version == 3;
  if (big) {
    if (a_lo) SVDD_GEMM256(2, false, 5 * 16 * 2 * 1024)     // three A stages + two W stages = 163,840 B (all of a CU's LDS); the epilogue's 8 wave slabs take 139,264 B of it
    else SVDD_GEMM256(1, false, 8 * 64 * 68 * 4)            // one-pass mode: the stages take 81,920 B, the epilogue slabs 139,264 B
    return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
  }
#undef SVDD_GEMM256
  const int mb = (M + G_BM - 1) / G_BM;
  const dim3 grid((unsigned)(((mb + 7) / 8) * 8 * (N / G_BN)));         // groups of 8 row tiles x N / 128 column tiles
  if (a_lo) hipLaunchKernelGGL(trunk_gemm_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(trunk_gemm_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_act_split(const float* x, const float* scale, const float* shift, int act, int64_t rows, int C,
                         int rows_per_seq, int pad, void* hi, void* lo, const int32_t* count, void* stream) {
  if (!x || !hi || rows <= 0 || C <= 0 || (C & 7) || rows_per_seq <= 0 || pad < 0 || 2 * pad > rows_per_seq || act < 0 || act > 2 ||
      ((scale == nullptr) != (shift == nullptr)))
    return SVDD_E_ARG;
  ActArgs a{x, scale, shift, act, rows, C, rows_per_seq, pad, (bf16_t*)hi, (bf16_t*)lo, count};
  const int64_t nthr = rows * (C >> 3);
  if (g_trunk_planes_f32) hipLaunchKernelGGL(trunk_act_split_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(trunk_act_split_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_layernorm_split(const float* x, const float* gamma, const float* beta, float eps, int64_t rows, int C,
                               void* hi, void* lo, const int32_t* count, int rows_per_seq, void* stream) {
  if (!x || !gamma || !beta || !hi || rows <= 0 || C <= 0 || (C & 7) || C > 4096 || (count && rows_per_seq <= 0)) return SVDD_E_ARG;
  LnArgs a{x, gamma, beta, eps, rows, C, (bf16_t*)hi, (bf16_t*)lo, count, rows_per_seq};
  if (g_trunk_planes_f32) hipLaunchKernelGGL(trunk_ln_split_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(trunk_ln_split_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_attn_pool(const float* x, const float* logits, int n, int L, int C, float* out, const int32_t* count,
                         void* out_hi, void* out_lo, const float* post_scale, const float* post_shift, int post_act, void* stream) {
  if (!x || !logits || (!out && !out_hi) || n <= 0 || L <= 0 || C <= 0 || (C & 3) || (out_lo && !out_hi) ||
      ((post_scale == nullptr) != (post_shift == nullptr)) || post_act < 0 || post_act > 2)
    return SVDD_E_ARG;
  PoolArgs a{x, logits, n, L, C, out, count, (bf16_t*)out_hi, (bf16_t*)out_lo, post_scale, post_shift, post_act};
  const int64_t nthr = (int64_t)n * ((L + 1) / 2 + 2) * (C >> 2);
  if (g_trunk_planes_f32) hipLaunchKernelGGL(trunk_attn_pool_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(trunk_attn_pool_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_attn_small(const float* qkv, const float* rel_k, const float* content_bias, const float* pos_bias, int n, int T,
                          int heads, int dk, int dv, void* hi, void* lo, const int32_t* count, void* stream) {
  if (!qkv || !rel_k || !content_bias || !pos_bias || !hi || n <= 0 || T < 1 || T > 4 || heads <= 0 || dk <= 0 || dv <= 0)
    return SVDD_E_ARG;
  AttnArgs a{qkv, rel_k, content_bias, pos_bias, n, heads, dk, dv, heads * (2 * dk + dv), 1.0f / sqrtf((float)dk),
             (bf16_t*)hi, (bf16_t*)lo, count};
  const dim3 grid((unsigned)(((int64_t)n * heads + 3) / 4));
#define SVDD_ATTN(T_)                                                                                                  \
  if (g_trunk_planes_f32) hipLaunchKernelGGL((trunk_attn_small_kernel<T_, true>), grid, dim3(256), 0, (hipStream_t)stream, a);  \
  else hipLaunchKernelGGL((trunk_attn_small_kernel<T_, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
  switch (T) {
    case 1: SVDD_ATTN(1) break;
    case 2: SVDD_ATTN(2) break;
    case 3: SVDD_ATTN(3) break;
    default: SVDD_ATTN(4) break;
  }
#undef SVDD_ATTN
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_stem_unfold(const uint8_t* tok, int n, int L, void* hi, const int32_t* count, void* stream) {
  if (!tok || !hi || n <= 0 || L <= 0) return SVDD_E_ARG;
  StemArgs a{tok, n, L, (bf16_t*)hi, count};
  const int64_t nthr = (int64_t)n * (L + 2) * 8;
  if (g_trunk_planes_f32) hipLaunchKernelGGL(trunk_stem_unfold_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(trunk_stem_unfold_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_windows(const uint8_t* cand, const uint8_t* parent, const int32_t* parent_idx, int div, int n, int L, int halo,
                       int depth, int slots, const int32_t* count, int32_t* w0, int32_t* wlen, int32_t* seg, void* stream) {
  if (!cand || !parent || !parent_idx || !w0 || !wlen || !seg || div <= 0 || n <= 0 || L <= 0 || L > 256 || halo < 0 || depth < 1 ||
      depth > 8 || (L & ((1 << (depth - 1)) - 1)) || slots < 1 || slots > WIN_K)
    return SVDD_E_ARG;
  WinArgs a{cand, parent, parent_idx, div, n, L, halo, depth, slots, count, w0, wlen, seg};
  hipLaunchKernelGGL(trunk_windows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_stem_unfold_win(const uint8_t* tok, int n, int L, int slots, const int32_t* w0, const int32_t* wlen,
                               const int32_t* off, void* hi, const int32_t* count, void* stream) {
  if (!tok || !hi || !w0 || !wlen || !off || n <= 0 || L <= 0 || slots < 1 || slots > WIN_K) return SVDD_E_ARG;
  StemWinArgs a{tok, n, L, slots, w0, wlen, off, (bf16_t*)hi, count};
  const int64_t nthr = (int64_t)n * L * 8;
  if (g_trunk_planes_f32) hipLaunchKernelGGL(trunk_stem_unfold_win_kernel<true>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(trunk_stem_unfold_win_kernel<false>, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_trunk_attn_pool_win(const float* x, const float* logits, int n, int L, int C, int in_halo, int slots, const int32_t* w0,
                             const int32_t* wlen, const int32_t* off, const int32_t* parent_idx, int div, const void* parent_hi,
                             const void* parent_lo, const int32_t* count, void* out_hi, void* out_lo, const float* post_scale,
                             const float* post_shift, int post_act, const int32_t* v0, const int32_t* vlen, const int32_t* off2,
                             void* stream) {
  if (!x || !logits || !w0 || !wlen || !off || !parent_idx || !parent_hi || !out_hi || n <= 0 || L <= 0 || ((L & 1) && v0) || C <= 0 ||
      (C & 3) || div <= 0 || in_halo < 0 || slots < 1 || slots > WIN_K || ((out_lo == nullptr) != (parent_lo == nullptr)) ||
      ((post_scale == nullptr) != (post_shift == nullptr)) || post_act < 0 || post_act > 2 ||
      ((v0 == nullptr) != (vlen == nullptr)) || ((v0 == nullptr) != (off2 == nullptr)))
    return SVDD_E_ARG;
  PoolWinArgs a{x, logits, n, L, C, in_halo, slots, w0, wlen, off, parent_idx, div, (const bf16_t*)parent_hi, (const bf16_t*)parent_lo,
                count, (bf16_t*)out_hi, (bf16_t*)out_lo, post_scale, post_shift, post_act, v0, vlen, off2};
  if (g_trunk_planes_f32) hipLaunchKernelGGL(trunk_attn_pool_win_kernel<true>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(trunk_attn_pool_win_kernel<false>, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

}  // extern "C"
