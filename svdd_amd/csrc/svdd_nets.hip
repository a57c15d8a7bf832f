// svdd_nets.hip — gfx950 kernels for the two networks of the SVDD step (the masked-diffusion backbone and the
// ConvGRU value net), which are >99 % of a diffusion step and which PyTorch-ROCm/MIOpen runs badly at SVDD's
// shapes (256 / 2560 short sequences, 64-128 channels). Exact fp32 on the matrix cores throughout. Exposed through
// the C ABI of include/svdd_hip.h ("net kernels" section); used by svdd_amd/fused.py.
//
//   backbone_kernel     the whole dilated-CNN backbone forward in one launch (residual stream in registers,
//                       LayerNorm'd activations in LDS, weights from L2 straight into MFMA B operands)
//   conv_tower_kernel   the value net's conv tower (stem + 5 residual blocks) in one launch, activations in LDS
//   gru_bidir_kernel    bidirectional GRU 64 -> 64, one launch for all L timesteps (MIOpen: ~2900 launches, 11.7 ms
//                       per value forward at n = 2560, L = 200 [rocprof r01_v0]); 16 sequences of one direction per
//                       workgroup, gate weights in registers as MFMA B operands, hidden state in LDS
//   value_tail_kernel   direction sum + LayerNorm + FFN dense1 + ReLU + (dense2 o head) + mean over length
//   conv1d_cl_*_kernel  one dilated conv layer (layer-wise path, small batches) ; epilogue_ln_kernel its epilogue
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "svdd_hip.h"
#include "svdd_spt.h"

extern "C" void svdd_internal_timed_events(int k, hipEvent_t* e0, hipEvent_t* e1);   // svdd_kernels.hip (profiling)

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}


constexpr int H = 64;          // hidden = input width
constexpr int TS = 16;         // sequences per workgroup (MFMA M)
constexpr int HPAD = H + 4;    // LDS row stride (floats): shifts rows by 16 B -> ds_read_b128 conflict-light

__device__ __forceinline__ float sigmoid_fast(float a) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * a));
}
__device__ __forceinline__ float tanh_fast(float a) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177793f * a));
}

// x      [n, L, 64] fp32 (channels-last conv-tower output)
// wpack  [2 dirs][4 waves][64 lanes][96] : lane (j = lane&15, g = lane>>4) of wave w holds, for
//        hidden unit u = 16w + j and input k = 16g + s (s = 0..15):
//        [0:16) W_ir[u][k]  [16:32) W_hr[u][k]  [32:48) W_iz  [48:64) W_hz  [64:80) W_in  [80:96) W_hn
// bpack  [2 dirs][4][64] : b_ir+b_hr, b_iz+b_hz, b_in, b_hn
// out    [2 dirs][n, L, 64] : hidden state of each direction at every timestep
// BOTH = false: one workgroup (4 waves) per (16-sequence tile, direction), grid (tiles, 2).
// BOTH = true : one workgroup (8 waves) per tile of `ts` <= 16 sequences runs BOTH directions, two waves per SIMD
//               interleaving their MFMAs; with ts = ceil(n / 256) every CU gets exactly one workgroup (n = 2560:
//               ts = 10), instead of 320 single-direction workgroups leaving 3/4 of the chip idle for the second
//               half of the launch.
template <bool BOTH>
__global__ __launch_bounds__(BOTH ? 512 : 256) void gru_bidir_kernel(const float* __restrict__ x, const float* __restrict__ wpack,
                                                                     const float* __restrict__ bpack, float* __restrict__ out,
                                                                     int n_alloc, int L, int ts, const int* __restrict__ count) {
  __shared__ __attribute__((aligned(16))) float hbuf_all[BOTH ? 2 : 1][2][TS][HPAD];
  const int lane = threadIdx.x & 63;
  const int w = (threadIdx.x >> 6) & 3;
  // BOTH = false: unit u = blockIdx.x covers (tile u >> 1, direction u & 1), so that the units of a compacted batch are a
  // dense prefix of the grid: the dispatcher hands consecutive workgroups to different CUs, and with the dead tiles
  // interleaved (grid (tiles, 2)) 256 live units of 320 landed two to a CU on 32 CUs — 678 us instead of 368.
  const int dir = BOTH ? (int)(threadIdx.x >> 8) : (int)(blockIdx.x & 1);
  float (*hbuf)[TS][HPAD] = hbuf_all[BOTH ? dir : 0];
  const int j = lane & 15;            // hidden unit within the wave's 16 / sequence row for A operands
  const int g = lane >> 4;            // k-group of A/B operands ; row-group of C/D
  const int seq0 = (BOTH ? (int)blockIdx.x : (int)(blockIdx.x >> 1)) * ts;
  // n_alloc: rows the output tensor is laid out for ; n: valid rows (device scalar when the batch was compacted)
  const int n = count ? __builtin_amdgcn_readfirstlane(*count) : n_alloc;
  if (seq0 >= n) return;

  // B operands: this lane's 96 weights stay in registers for the whole sequence
  float wr[96];
  {
    const float4* wp = reinterpret_cast<const float4*>(wpack + (((size_t)dir * 4 + w) * 64 + lane) * 96);
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const float4 v = wp[i];
      wr[4 * i] = v.x; wr[4 * i + 1] = v.y; wr[4 * i + 2] = v.z; wr[4 * i + 3] = v.w;
    }
  }
  const int u = 16 * w + j;
  const float b_r = bpack[(dir * 4 + 0) * H + u], b_z = bpack[(dir * 4 + 1) * H + u];
  const float b_nx = bpack[(dir * 4 + 2) * H + u], b_nh = bpack[(dir * 4 + 3) * H + u];

  // A-operand source row for this lane (clamped for the ragged last tile)
  const int arow = min(seq0 + min(j, ts - 1), n - 1);
  const float* xrow = x + (size_t)arow * L * H + 16 * g;
  float hprev[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // h[seq = 4g + rho][u] of the previous step

  // h_0 = 0
  for (int i = threadIdx.x & 255; i < TS * HPAD; i += 256) (&hbuf[0][0][0])[i] = 0.0f;

  const int t0 = dir == 0 ? 0 : L - 1;
  const int dt = dir == 0 ? 1 : -1;
  float xa[16], xn[16];
  {
    const float4* xp = reinterpret_cast<const float4*>(xrow + (size_t)t0 * H);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = xp[i]; xa[4 * i] = v.x; xa[4 * i + 1] = v.y; xa[4 * i + 2] = v.z; xa[4 * i + 3] = v.w; }
  }
  // input projections of step 0 (they do not depend on h): acc = bias + x_0 W_i*
  f32x4 acc_r = {b_r, b_r, b_r, b_r}, acc_z = {b_z, b_z, b_z, b_z}, acc_nx = {b_nx, b_nx, b_nx, b_nx};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[s], acc_r, 0, 0, 0);
    acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[32 + s], acc_z, 0, 0, 0);
    acc_nx = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[64 + s], acc_nx, 0, 0, 0);
  }
  __syncthreads();

  // Per step the only serial chain is  barrier -> read h -> 48 recurrent MFMAs -> gates -> write h.
  // The 48 input-projection MFMAs of the NEXT step are issued between the h write and the barrier, so they cover
  // the barrier skew and the LDS write latency instead of adding to the chain.
  for (int step = 0; step < L; ++step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    if (step + 1 < L) {             // prefetch x_{t+1}
      const float4* xp = reinterpret_cast<const float4*>(xrow + (size_t)(t + dt) * H);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = xp[i]; xn[4 * i] = v.x; xn[4 * i + 1] = v.y; xn[4 * i + 2] = v.z; xn[4 * i + 3] = v.w; }
    }
    float ha[16];
    {
      const float4* hp = reinterpret_cast<const float4*>(&hbuf[cur][j][16 * g]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = hp[i]; ha[4 * i] = v.x; ha[4 * i + 1] = v.y; ha[4 * i + 2] = v.z; ha[4 * i + 3] = v.w; }
    }
    f32x4 acc_nh = {b_nh, b_nh, b_nh, b_nh};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc_nh = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[80 + s], acc_nh, 0, 0, 0);
      acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[16 + s], acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[48 + s], acc_z, 0, 0, 0);
    }
    // C/D layout: reg rho -> row (sequence) 4g + rho, column (unit) j
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const float r = sigmoid_fast(acc_r[rho]);
      const float z = sigmoid_fast(acc_z[rho]);
      const float nn = tanh_fast(acc_nx[rho] + r * acc_nh[rho]);
      const float hn = (1.0f - z) * nn + z * hprev[rho];
      hprev[rho] = hn;
      const int srow = 4 * g + rho;
      hbuf[cur ^ 1][srow][u] = hn;
      if (srow < ts && seq0 + srow < n) out[(((size_t)dir * n_alloc + seq0 + srow) * L + t) * H + u] = hn;
    }
    // next step's input projections (independent of the barrier below)
    acc_r = f32x4{b_r, b_r, b_r, b_r}; acc_z = f32x4{b_z, b_z, b_z, b_z}; acc_nx = f32x4{b_nx, b_nx, b_nx, b_nx};
    if (step + 1 < L) {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(xn[s], wr[s], acc_r, 0, 0, 0);
        acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(xn[s], wr[32 + s], acc_z, 0, 0, 0);
        acc_nx = __builtin_amdgcn_mfma_f32_16x16x4f32(xn[s], wr[64 + s], acc_nx, 0, 0, 0);
      }
    }
    __syncthreads();
  }
}

// Producer / consumer variant of gru_bidir_kernel<false> (same bits: every accumulator sees the same products in the same
// order). Workgroup = 8 waves on one (tile of 16 sequences, direction): waves 0-3 are CONSUMERS and run the recurrence
// (read h_t and the input projections of step t from LDS -> 48 recurrent MFMAs -> gates -> write h_{t+1}); waves 4-7
// are PRODUCERS and compute bias + W_i* x of step t + 1 into an LDS ring one step ahead, from x rows they prefetched two
// further steps ahead. Consumer w and producer w + 4 share a SIMD and own the same 16 hidden units, so a producer lane's
// accumulators are exactly what its partner consumer lane continues from. The consumer's serial chain loses the 48
// input-projection MFMAs and the x loads. Measured (tools/gru_microbench.py): 369 -> 359 us at <= 256 units, 727 -> 670 us
// at n = 2560 — the step stays at ~1.8 us against the 1.28 us the 96 MFMAs of a wave pair need; wave priorities
// (s_setprio 3 for the consumers) changed nothing.
// (The 16-bit twin gru_lp_kernel, svdd_lp_gru_tail.hip, was built this way first.)
__global__ __launch_bounds__(512) void gru_pc_kernel(const float* __restrict__ x, const float* __restrict__ wpack,
                                                     const float* __restrict__ bpack, float* __restrict__ out,
                                                     int n_alloc, int L, const int* __restrict__ count) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][TS][HPAD];
  __shared__ __attribute__((aligned(16))) float xproj[2][4][3][64 * 4];        // [slot][wave][gate][lane x 4]
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool producer = wv >= 4;
  const int w = wv & 3;
  const int dir = (int)(blockIdx.x & 1);               // unit u = (tile u >> 1, direction u & 1): dense prefix when compacted
  const int j = lane & 15, g = lane >> 4;
  const int seq0 = (int)(blockIdx.x >> 1) * TS;
  const int n = count ? __builtin_amdgcn_readfirstlane(*count) : n_alloc;
  if (seq0 >= n) return;
  const int u = 16 * w + j;
  const int t0 = dir == 0 ? 0 : L - 1;
  const int dt = dir == 0 ? 1 : -1;

  // this wave's half of the lane's 96 gate weights: producers W_ir, W_iz, W_in ; consumers W_hr, W_hz, W_hn
  float wr[3][16];
  {
    const float4* wp = reinterpret_cast<const float4*>(wpack + (((size_t)dir * 4 + w) * 64 + lane) * 96 + (producer ? 0 : 16));
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = wp[8 * m + i];
        wr[m][4 * i] = v.x; wr[m][4 * i + 1] = v.y; wr[m][4 * i + 2] = v.z; wr[m][4 * i + 3] = v.w;
      }
  }
  for (int i = threadIdx.x; i < TS * HPAD; i += 512) (&hbuf[0][0][0])[i] = 0.0f;          // h_0 = 0
  float* myproj = &xproj[0][w][0][lane * 4];
  constexpr int SLOT = 4 * 3 * 64 * 4;                 // floats per ring slot

  if (producer) {
    const float b_r = bpack[(dir * 4 + 0) * H + u], b_z = bpack[(dir * 4 + 1) * H + u], b_nx = bpack[(dir * 4 + 2) * H + u];
    const int arow = min(seq0 + j, n - 1);             // clamped for the ragged last tile
    const float* xrow = x + (size_t)arow * L * H + 16 * g;
    float xa[2][16];                                   // two x rows in flight
    auto load_x = [&](int t, int bufi) {
      const float4* xp = reinterpret_cast<const float4*>(xrow + (size_t)t * H);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = xp[i];
        xa[bufi][4 * i] = v.x; xa[bufi][4 * i + 1] = v.y; xa[bufi][4 * i + 2] = v.z; xa[bufi][4 * i + 3] = v.w;
      }
    };
    auto project = [&](int bufi, int slot) {           // bias + x W_i* of one step -> ring slot
      f32x4 pr = {b_r, b_r, b_r, b_r}, pz = {b_z, b_z, b_z, b_z}, pn = {b_nx, b_nx, b_nx, b_nx};
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        pr = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[bufi][s], wr[0][s], pr, 0, 0, 0);
        pz = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[bufi][s], wr[1][s], pz, 0, 0, 0);
        pn = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[bufi][s], wr[2][s], pn, 0, 0, 0);
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
    // (the prefetches are unconditional — past the end they re-read the last row: with a load on one path only, the compiler's
    //  s_waitcnt before a projection became vmcnt(0) and drained the row issued just before it, not only the one it needs)
    for (int s = 0; s < L; s += 2) {
      if (s + 1 < L) project(1, 1);
      load_x(t0 + min(s + 3, L - 1) * dt, 1);
      __syncthreads();
      if (s + 1 >= L) break;
      if (s + 2 < L) project(0, 0);
      load_x(t0 + min(s + 4, L - 1) * dt, 0);
      __syncthreads();
    }
    return;
  }

  // ---- consumers
  const float b_nh = bpack[(dir * 4 + 3) * H + u];
  float hprev[4] = {0.0f, 0.0f, 0.0f, 0.0f};           // h[seq = 4g + rho][u] of the previous step
  __syncthreads();                                     // h_0 and slot 0 are ready
  for (int step = 0; step < L; ++step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    float ha[16];
    {
      const float4* hp = reinterpret_cast<const float4*>(&hbuf[cur][j][16 * g]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = hp[i]; ha[4 * i] = v.x; ha[4 * i + 1] = v.y; ha[4 * i + 2] = v.z; ha[4 * i + 3] = v.w; }
    }
    const float* src = myproj + cur * SLOT;
    f32x4 acc_r = *reinterpret_cast<const f32x4*>(src);
    f32x4 acc_z = *reinterpret_cast<const f32x4*>(src + 256);
    const f32x4 acc_nx = *reinterpret_cast<const f32x4*>(src + 512);
    f32x4 acc_nh = {b_nh, b_nh, b_nh, b_nh};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc_nh = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[2][s], acc_nh, 0, 0, 0);
      acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[0][s], acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wr[1][s], acc_z, 0, 0, 0);
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {                // C/D layout: reg rho -> row (sequence) 4g + rho, column (unit) j
      const float r = sigmoid_fast(acc_r[rho]);
      const float z = sigmoid_fast(acc_z[rho]);
      const float nn = tanh_fast(acc_nx[rho] + r * acc_nh[rho]);
      const float hn = (1.0f - z) * nn + z * hprev[rho];
      hprev[rho] = hn;
      const int srow = 4 * g + rho;
      hbuf[cur ^ 1][srow][u] = hn;
      if (seq0 + srow < n) out[(((size_t)dir * n_alloc + seq0 + srow) * L + t) * H + u] = hn;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------ fused conv epilogue + LayerNorm ----
// Channels-last rows [R, C]. One pass replaces the bias-add, ReLU, residual-add, time-bias-add and
// LayerNorm kernels PyTorch launches between two convolutions of the dilated-CNN backbone
// (reference models/dnaconv.py:188-197) and of the value net's conv tower (Enformer.py:2269-2285):
//     t     = y + bias
//     f_out = act 0: relu(t) + f_prev | act 1: relu(t + f_prev) | act 2: t + f_prev      (f_prev optional)
//     hn    = LayerNorm(f_out + tb) * gamma + beta                    (optional; eps = 1e-5, biased variance)
// One wave per row, lane owns VPL = C/64 consecutive channels (8-16 B vector loads), mean/variance by
// xor-shuffle wave reduction (two-pass, like ATen's RowwiseMoments). HBM-bound: 2 reads + 1-2 writes.
struct EpiArgs {
  const float* y; const float* bias; const float* f_prev; const float* tb; const float* gamma; const float* beta;
  float* f_out; float* hn; int64_t R; int act;
};


template <int VPL>
__global__ __launch_bounds__(256) void epilogue_ln_kernel(EpiArgs a) {
  constexpr int C = 64 * VPL;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const int c0 = lane * VPL;
  float bias[VPL], tb[VPL], gm[VPL], bt[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    bias[i] = a.bias ? a.bias[c0 + i] : 0.0f;
    tb[i] = (a.hn && a.tb) ? a.tb[c0 + i] : 0.0f;
    gm[i] = a.hn ? a.gamma[c0 + i] : 1.0f;
    bt[i] = a.hn ? a.beta[c0 + i] : 0.0f;
  }
  for (int64_t r = wave; r < a.R; r += nwaves) {
    const int64_t base = r * C + c0;
    float v[VPL], p[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { v[i] = a.y[base + i]; p[i] = a.f_prev ? a.f_prev[base + i] : 0.0f; }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float t = v[i] + bias[i];
      v[i] = a.act == 0 ? fmaxf(t, 0.0f) + p[i] : a.act == 1 ? fmaxf(t + p[i], 0.0f) : t + p[i];
      if (a.f_out) a.f_out[base + i] = v[i];
      v[i] += tb[i];
      s += v[i];
    }
    if (a.hn) {
      const float mean = wave_sum(s) * (1.0f / C);
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) { const float d = v[i] - mean; q += d * d; }
      const float rstd = rsqrtf(wave_sum(q) * (1.0f / C) + 1e-5f);
#pragma unroll
      for (int i = 0; i < VPL; ++i) a.hn[base + i] = (v[i] - mean) * rstd * gm[i] + bt[i];
    }
  }
}

// ------------------------------------------------------------- dilated Conv1d, channels-last, fp32 MFMA ----
// y[n, l, co] = sum_{t, ci} x[n, l + (t - T/2) * dil, ci] * W[co, ci, t]      ("same" zero padding, no bias)
// as an implicit GEMM on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32). One workgroup owns a tile
// of whole sequences (<= 224 rows: one L=200 sequence, four L=50 sequences), so that
//   * the activations of the tile are staged into LDS ONCE per 32-channel chunk and re-used by all T taps
//     (a tap is just a row offset into the LDS image; out-of-range rows read a zero row);
//   * taps whose receptive rows are all padding are skipped, per 32-row MFMA tile — at L=200 the
//     dilation-64 layers only do 35 % of the nominal work, dilation-16 82 %;
//   * B = 256 sequences is exactly one workgroup per CU (no tail).
// Wave w owns output channels [32 (w % NCT), +32) and row tiles w / NCT, w / NCT + 4 / NCT, ...;
// accumulators stay in registers across the whole K loop (Cin * T).
constexpr int CONV_RT = 7;                 // 32-row MFMA tiles per workgroup tile
constexpr int CONV_ROWS = 32 * CONV_RT;    // 224
constexpr int CH = 32;                     // input-channel chunk
constexpr int CHP = CH + 4;                // padded LDS row stride (floats): conflict-free ds_read_b128

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x; const float* wpack; float* y;
  int n, L, spt /*sequences per tile*/, T, dil;
  const float* bias; const float* f_prev; int act;      // fused epilogue (static kernels): act -1 = raw conv output
  const float* tb; const float* gamma; const float* beta; float* hn;   // + LayerNorm(y + tb) of the next layer (hn != NULL)
};

// 16 MFMAs of one 32-row tile: A = 4 float4 (16 consecutive input channels of this lane's row), B = bf[16]
#define CONV_MMA16(ACC, A0, A1, A2, A3)                                                    \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.x, bf[0], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.y, bf[1], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.z, bf[2], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A0.w, bf[3], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.x, bf[4], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.y, bf[5], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.z, bf[6], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A1.w, bf[7], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A2.x, bf[8], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A2.y, bf[9], ACC, 0, 0, 0);                   \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A2.z, bf[10], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A2.w, bf[11], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A3.x, bf[12], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A3.y, bf[13], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A3.z, bf[14], ACC, 0, 0, 0);                  \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A3.w, bf[15], ACC, 0, 0, 0);

// two row tiles interleaved: consecutive MFMAs never share an accumulator (a dependent back-to-back pair
// stalls ~43 cycles as soon as ANY other instruction is scheduled between them)
#define CONV_MMA2(K, AV, BV)                                                               \
  acc[ra] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, bf[K], acc[ra], 0, 0, 0);            \
  acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(BV, bf[K], acc[rb], 0, 0, 0);
#define CONV_MMA32(X0, X1, X2, X3, Y0, Y1, Y2, Y3)                                         \
  CONV_MMA2(0, X0.x, Y0.x) CONV_MMA2(1, X0.y, Y0.y) CONV_MMA2(2, X0.z, Y0.z) CONV_MMA2(3, X0.w, Y0.w)       \
  CONV_MMA2(4, X1.x, Y1.x) CONV_MMA2(5, X1.y, Y1.y) CONV_MMA2(6, X1.z, Y1.z) CONV_MMA2(7, X1.w, Y1.w)       \
  CONV_MMA2(8, X2.x, Y2.x) CONV_MMA2(9, X2.y, Y2.y) CONV_MMA2(10, X2.z, Y2.z) CONV_MMA2(11, X2.w, Y2.w)     \
  CONV_MMA2(12, X3.x, Y3.x) CONV_MMA2(13, X3.y, Y3.y) CONV_MMA2(14, X3.z, Y3.z) CONV_MMA2(15, X3.w, Y3.w)

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void conv1d_cl_kernel(ConvArgs a) {
  constexpr int NCHUNK = CIN / CH;
  constexpr int NCT = COUT / 32;                 // column tiles (2 or 4)
  constexpr int RGROUPS = 4 / NCT;               // waves sharing a column tile split the row tiles
  constexpr int MAXRT = (CONV_RT + RGROUPS - 1) / RGROUPS;
  constexpr int BLD = COUT * 8 / 256;            // float4 per thread to stage one W[t][c] tile (2 or 4)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                      // [CONV_ROWS + 1][CHP]   (+1: the zero row)
  float* Bs = smem + (CONV_ROWS + 1) * CHP;              // [2][COUT][CHP]

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform (SGPR): keeps the tile-liveness branches scalar
  const int i = lane & 31, h = lane >> 5;
  const int ct = w % NCT, rg = w / NCT;
  const int L = a.L, T = a.T;
  const int tile_rows = a.spt * L;
  const int64_t row0 = (int64_t)blockIdx.x * tile_rows;   // first flattened (n*L) row of this tile
  const int64_t total_rows = (int64_t)a.n * L;
  const int half_t = T / 2;
  const int NIT = NCHUNK * T;

  f32x16 acc[MAXRT];
#pragma unroll
  for (int r = 0; r < MAXRT; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.0f;

  // this lane's A row per owned row tile: LDS byte-free offsets and the position inside its sequence
  int arow[MAXRT], apos[MAXRT];
#pragma unroll
  for (int r = 0; r < MAXRT; ++r) {
    const int rt = rg + r * RGROUPS;
    const int tr = 32 * rt + i;
    arow[r] = tr;
    apos[r] = (rt < CONV_RT && tr < tile_rows) ? tr % L : -(1 << 20);   // padding rows never match a tap
  }
  for (int e = threadIdx.x; e < CHP; e += 256) As[CONV_ROWS * CHP + e] = 0.0f;   // zero row

  // staging coordinates of this thread for the weight tile W[t][c] ([COUT][32] contiguous in wpack)
  const int b_r = threadIdx.x >> 3, b_q = threadIdx.x & 7;             // rows b_r + 32 k, 16-B column b_q
  float4 bp0, bp1, bp2, bp3;                                            // prefetched tile (named: must stay in VGPRs)
  bp0 = bp1 = bp2 = bp3 = float4{0.0f, 0.0f, 0.0f, 0.0f};

  // first live tap
  int it = 0;
  while (it < NIT) { const int d = (it % T - half_t) * a.dil; if (max(0, -d) < min(L, L - d)) break; ++it; }
  if (it < NIT) {
    const float* wsrc = a.wpack + ((size_t)((it % T) * NCHUNK + it / T) * COUT) * CH + b_r * CH + 4 * b_q;
    bp0 = *reinterpret_cast<const float4*>(wsrc);
    bp1 = *reinterpret_cast<const float4*>(wsrc + 32 * CH);
    if (BLD > 2) { bp2 = *reinterpret_cast<const float4*>(wsrc + 64 * CH); bp3 = *reinterpret_cast<const float4*>(wsrc + 96 * CH); }
    float* Bt = Bs + b_r * CHP + 4 * b_q;
    *reinterpret_cast<float4*>(Bt) = bp0;
    *reinterpret_cast<float4*>(Bt + 32 * CHP) = bp1;
    if (BLD > 2) { *reinterpret_cast<float4*>(Bt + 64 * CHP) = bp2; *reinterpret_cast<float4*>(Bt + 96 * CHP) = bp3; }
  }
  int par = 0, cur_c = -1;
  while (it < NIT) {
    const int c = it / T, t = it % T;
    if (c != cur_c) {                                        // new input-channel chunk: restage the activations
      __syncthreads();                                       // previous chunk's readers are done with As
      for (int e = threadIdx.x; e < CONV_ROWS * 8; e += 256) {
        const int r = e >> 3, q = e & 7;
        float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (r < tile_rows && row0 + r < total_rows)
          v = *reinterpret_cast<const float4*>(a.x + (row0 + r) * CIN + c * CH + 4 * q);
        *reinterpret_cast<float4*>(As + r * CHP + 4 * q) = v;
      }
      cur_c = c;
    }
    // next live tap; its weight tile is fetched now and flies under this tap's MFMAs
    int nxt = it + 1;
    while (nxt < NIT) { const int d = (nxt % T - half_t) * a.dil; if (max(0, -d) < min(L, L - d)) break; ++nxt; }
    if (nxt < NIT) {
      const float* wsrc = a.wpack + ((size_t)((nxt % T) * NCHUNK + nxt / T) * COUT) * CH + b_r * CH + 4 * b_q;
      bp0 = *reinterpret_cast<const float4*>(wsrc);
      bp1 = *reinterpret_cast<const float4*>(wsrc + 32 * CH);
      if (BLD > 2) { bp2 = *reinterpret_cast<const float4*>(wsrc + 64 * CH); bp3 = *reinterpret_cast<const float4*>(wsrc + 96 * CH); }
    }
    __syncthreads();                                         // Bs[par] (and As) visible to every wave
    float bf[16];
    {
      const float4* bq = reinterpret_cast<const float4*>(Bs + par * COUT * CHP + (32 * ct + i) * CHP + 16 * h);
      const float4 v0 = bq[0], v1 = bq[1], v2 = bq[2], v3 = bq[3];
      bf[0] = v0.x; bf[1] = v0.y; bf[2] = v0.z; bf[3] = v0.w; bf[4] = v1.x; bf[5] = v1.y; bf[6] = v1.z; bf[7] = v1.w;
      bf[8] = v2.x; bf[9] = v2.y; bf[10] = v2.z; bf[11] = v2.w; bf[12] = v3.x; bf[13] = v3.y; bf[14] = v3.z; bf[15] = v3.w;
    }
    const int delta = (t - half_t) * a.dil;
    const int lo = max(0, -delta), hi = min(L, L - delta);   // rows l with 0 <= l + delta < L are live
    const float* Ah = As + 16 * h;
    // software pipeline over the row tiles: the 4 ds_read_b128 of tile r+1 are issued before the 16 MFMAs of tile r
    float4 x0, x1, x2, x3, y0, y1, y2, y3;
    x0 = x1 = x2 = x3 = y0 = y1 = y2 = y3 = float4{0.0f, 0.0f, 0.0f, 0.0f};
#define CONV_LIVE(R) ((rg + (R) * RGROUPS) < CONV_RT && (a.spt == 1 ? (lo < 32 * (rg + (R) * RGROUPS) + 32 && hi > 32 * (rg + (R) * RGROUPS)) \
                                                                      : 32 * (rg + (R) * RGROUPS) < tile_rows))
#define CONV_ALOAD(R, V0, V1, V2, V3)                                                                    \
  {                                                                                                       \
    const int p_ = apos[R] + delta;                                                                       \
    const float4* ap_ = reinterpret_cast<const float4*>(Ah + ((p_ >= 0 && p_ < L) ? arow[R] + delta : CONV_ROWS) * CHP); \
    V0 = ap_[0]; V1 = ap_[1]; V2 = ap_[2]; V3 = ap_[3];                                                    \
  }
    // row tiles in pairs (2 independent accumulator chains interleaved); dead tiles of a dilated tap are skipped
#pragma unroll
    for (int r = 0; r < MAXRT; r += 2) {
      const int ra = r, rb = (r + 1 < MAXRT) ? r + 1 : r;
      const bool la = CONV_LIVE(ra), lb = (r + 1 < MAXRT) && CONV_LIVE(rb);
      if (la && lb) {
        CONV_ALOAD(ra, x0, x1, x2, x3)
        CONV_ALOAD(rb, y0, y1, y2, y3)
        CONV_MMA32(x0, x1, x2, x3, y0, y1, y2, y3)
      } else if (la) {
        CONV_ALOAD(ra, x0, x1, x2, x3)
        CONV_MMA16(acc[ra], x0, x1, x2, x3)
      } else if (lb) {
        CONV_ALOAD(rb, y0, y1, y2, y3)
        CONV_MMA16(acc[rb], y0, y1, y2, y3)
      }
    }
    if (nxt < NIT) {                                        // other buffer: its last readers passed this tap's barrier
      float* Bt = Bs + (par ^ 1) * COUT * CHP + b_r * CHP + 4 * b_q;
      *reinterpret_cast<float4*>(Bt) = bp0;
      *reinterpret_cast<float4*>(Bt + 32 * CHP) = bp1;
      if (BLD > 2) { *reinterpret_cast<float4*>(Bt + 64 * CHP) = bp2; *reinterpret_cast<float4*>(Bt + 96 * CHP) = bp3; }
    }
    par ^= 1;
    it = nxt;
  }
#undef CONV_LIVE
#undef CONV_ALOAD
  // C/D layout of 32x32: reg e -> row (e & 3) + 8 (e >> 2) + 4 h, column i
#pragma unroll
  for (int r = 0; r < MAXRT; ++r) {
    const int rt = rg + r * RGROUPS;
    if (rt >= CONV_RT) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int tr = 32 * rt + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (tr < tile_rows && row0 + tr < total_rows) a.y[(row0 + tr) * COUT + 32 * ct + i] = acc[r][e];
    }
  }
}

// ---- statically scheduled variant: taps, dilation and sequence length are template parameters, so which
// (tap, 32-row tile) pairs touch only padding is known at compile time: no branch surrounds any MFMA, the
// accumulators stay in place, and the compiler software-pipelines the straight-line tap bodies.
constexpr bool conv_tap_live(int t, int T, int dil, int L) {
  const int d = (t - T / 2) * dil;
  const int lo = d < 0 ? -d : 0, hi = d > 0 ? L - d : L;
  return lo < hi;
}
constexpr bool conv_tile_live(int t, int rt, int T, int dil, int L, int spt) {
  const int d = (t - T / 2) * dil;
  const int lo = d < 0 ? -d : 0, hi = d > 0 ? L - d : L;
  if (lo >= hi) return false;
  if (32 * rt >= spt * L) return false;                       // pure padding tile
  if (spt != 1) return true;
  return lo < 32 * rt + 32 && hi > 32 * rt;
}

template <int CIN, int COUT, int T, int DIL, int LSEQ>
__global__ __launch_bounds__(256) void conv1d_cl_static_kernel(ConvArgs a) {
  constexpr int NCHUNK = CIN / CH;
  constexpr int NCT = COUT / 32;                 // 4 (128 channels) or 2 (64)
  constexpr int RGROUPS = 4 / NCT;               // waves sharing a column tile take interleaved row tiles
  constexpr int NRT = (CONV_RT + RGROUPS - 1) / RGROUPS;   // row tiles per wave (the last one of group 1 is padding)
  constexpr int BLD = COUT * 8 / 256;            // float4 per thread per weight tile
  constexpr int SPT = CONV_ROWS / LSEQ;
  constexpr int TILE_ROWS = SPT * LSEQ;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                      // [CONV_ROWS + 1][CHP]
  float* Bs = smem + (CONV_ROWS + 1) * CHP;              // [2][COUT][CHP]

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ct = w % NCT, rg = w / NCT;                  // this wave's 32 output channels / row-tile group
  const int i = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * TILE_ROWS;
  const int64_t total_rows = (int64_t)a.n * LSEQ;

  f32x16 acc[NRT];
#pragma unroll
  for (int r = 0; r < NRT; ++r)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[r][e] = 0.0f;

  // row tile r of this wave is tile rt = rg + r * RGROUPS; arow = its row for this lane, apos = position in its sequence
  int arow[NRT], apos[NRT];
#pragma unroll
  for (int r = 0; r < NRT; ++r) {
    arow[r] = 32 * (rg + r * RGROUPS) + i;
    apos[r] = (arow[r] < TILE_ROWS) ? arow[r] % LSEQ : -(1 << 20);      // padding rows read the zero row
  }
  for (int e = threadIdx.x; e < CHP; e += 256) As[CONV_ROWS * CHP + e] = 0.0f;   // zero row

  const int b_r = threadIdx.x >> 3, b_q = threadIdx.x & 7;
  const float* wbase = a.wpack + b_r * CH + 4 * b_q;
  float* bdst = Bs + b_r * CHP + 4 * b_q;
  float4 bp0, bp1, bp2, bp3;
  bp2 = bp3 = float4{0.0f, 0.0f, 0.0f, 0.0f};
  int par = 0;
  {                                                      // first live tap of chunk 0
    int t0 = 0;
#pragma unroll
    for (int t = T - 1; t >= 0; --t) if (conv_tap_live(t, T, DIL, LSEQ)) t0 = t;
    const float* wsrc = wbase + ((size_t)(t0 * NCHUNK) * COUT) * CH;
    bp0 = *reinterpret_cast<const float4*>(wsrc);
    bp1 = *reinterpret_cast<const float4*>(wsrc + 32 * CH);
    if (BLD > 2) { bp2 = *reinterpret_cast<const float4*>(wsrc + 64 * CH); bp3 = *reinterpret_cast<const float4*>(wsrc + 96 * CH); }
    *reinterpret_cast<float4*>(bdst) = bp0;
    *reinterpret_cast<float4*>(bdst + 32 * CHP) = bp1;
    if (BLD > 2) { *reinterpret_cast<float4*>(bdst + 64 * CHP) = bp2; *reinterpret_cast<float4*>(bdst + 96 * CHP) = bp3; }
  }
  const float* Ah = As + 16 * h;
  for (int c = 0; c < NCHUNK; ++c) {
    __syncthreads();                                       // previous chunk's readers are done with As
    for (int e = threadIdx.x; e < CONV_ROWS * 8; e += 256) {
      const int r = e >> 3, q = e & 7;
      float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (r < TILE_ROWS && row0 + r < total_rows)
        v = *reinterpret_cast<const float4*>(a.x + (row0 + r) * CIN + c * CH + 4 * q);
      *reinterpret_cast<float4*>(As + r * CHP + 4 * q) = v;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (!conv_tap_live(t, T, DIL, LSEQ)) continue;       // compile-time
      // weight tile of the next live tap (next chunk's first live tap after the last one) flies under the MFMAs
      int tn = -1;
#pragma unroll
      for (int u = T - 1; u > t; --u) if (conv_tap_live(u, T, DIL, LSEQ)) tn = u;
      int cn = c;
      if (tn < 0) {
        cn = c + 1;
#pragma unroll
        for (int u = T - 1; u >= 0; --u) if (conv_tap_live(u, T, DIL, LSEQ)) tn = u;
      }
      const bool have_next = cn < NCHUNK;
      if (have_next) {
        const float* wsrc = wbase + ((size_t)(tn * NCHUNK + cn) * COUT) * CH;
        bp0 = *reinterpret_cast<const float4*>(wsrc);
        bp1 = *reinterpret_cast<const float4*>(wsrc + 32 * CH);
        if (BLD > 2) { bp2 = *reinterpret_cast<const float4*>(wsrc + 64 * CH); bp3 = *reinterpret_cast<const float4*>(wsrc + 96 * CH); }
      }
      __syncthreads();                                     // Bs[par] (and As) visible to every wave
      float bf[16];
      {
        const float4* bq = reinterpret_cast<const float4*>(Bs + par * COUT * CHP + (32 * ct + i) * CHP + 16 * h);
        const float4 v0 = bq[0], v1 = bq[1], v2 = bq[2], v3 = bq[3];
        bf[0] = v0.x; bf[1] = v0.y; bf[2] = v0.z; bf[3] = v0.w; bf[4] = v1.x; bf[5] = v1.y; bf[6] = v1.z; bf[7] = v1.w;
        bf[8] = v2.x; bf[9] = v2.y; bf[10] = v2.z; bf[11] = v2.w; bf[12] = v3.x; bf[13] = v3.y; bf[14] = v3.z; bf[15] = v3.w;
      }
      constexpr int dummy = 0; (void)dummy;
      const int delta = (t - T / 2) * DIL;
      // Tile liveness is a compile-time fact when every wave owns all row tiles (RGROUPS == 1); with two row
      // groups a tile's index depends on the wave, so only whole-tap skipping is static there.
#define SCONV_LIVE(R) (RGROUPS == 1 ? conv_tile_live(t, R, T, DIL, LSEQ, SPT) : true)
      float4 af[NRT][4];
#pragma unroll
      for (int r = 0; r < NRT; ++r) {
        if (!SCONV_LIVE(r)) continue;
        const int p = apos[r] + delta;
        const float4* ap = reinterpret_cast<const float4*>(Ah + ((p >= 0 && p < LSEQ) ? arow[r] + delta : CONV_ROWS) * CHP);
        af[r][0] = ap[0]; af[r][1] = ap[1]; af[r][2] = ap[2]; af[r][3] = ap[3];
      }
      // k-step outer, row tile inner: consecutive MFMAs never share an accumulator
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int r = 0; r < NRT; ++r)
          if (SCONV_LIVE(r)) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][q].x, bf[4 * q], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRT; ++r)
          if (SCONV_LIVE(r)) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][q].y, bf[4 * q + 1], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRT; ++r)
          if (SCONV_LIVE(r)) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][q].z, bf[4 * q + 2], acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NRT; ++r)
          if (SCONV_LIVE(r)) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[r][q].w, bf[4 * q + 3], acc[r], 0, 0, 0);
      }
#undef SCONV_LIVE
      if (have_next) {                                    // other buffer: its last readers passed this tap's barrier
        float* bd = bdst + (par ^ 1) * COUT * CHP;
        *reinterpret_cast<float4*>(bd) = bp0;
        *reinterpret_cast<float4*>(bd + 32 * CHP) = bp1;
        if (BLD > 2) { *reinterpret_cast<float4*>(bd + 64 * CHP) = bp2; *reinterpret_cast<float4*>(bd + 96 * CHP) = bp3; }
      }
      par ^= 1;
    }
  }
  // ---- epilogue ----
  if (a.act < 0) {                                        // raw conv output straight from the accumulators
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int tr = 32 * (rg + r * RGROUPS) + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (tr < TILE_ROWS && row0 + tr < total_rows) a.y[(row0 + tr) * COUT + 32 * ct + i] = acc[r][e];
      }
    }
    return;
  }
  // Fused: the accumulator tile goes through LDS (two halves of 128 rows: it does not fit at once) so that the
  // residual read, the f store and the LayerNorm run row-wise with fully coalesced 256/512-B rows:
  //   t = conv + bias ; f = act 0: relu(t) + f_prev | 1: relu(t + f_prev) | 2: t + f_prev ; hn = LN(f + tb)*gamma + beta
  constexpr int EP = COUT + 4;                            // LDS row stride of the epilogue tile
  constexpr int VPL = COUT / 64;                          // channels per lane in the row-wise pass
  float* Es = smem;
  const float bias_c = a.bias ? a.bias[32 * ct + i] : 0.0f;
  float tbv[VPL], gmv[VPL], btv[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    tbv[k] = (a.hn && a.tb) ? a.tb[lane * VPL + k] : 0.0f;
    gmv[k] = a.hn ? a.gamma[lane * VPL + k] : 1.0f;
    btv[k] = a.hn ? a.beta[lane * VPL + k] : 0.0f;
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();                                      // main loop / previous half is done with this LDS
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
      const int rt = rg + r * RGROUPS;
      if (rt / 4 != half || rt >= CONV_RT) continue;      // scalar-uniform
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int lr = 32 * (rt - 4 * half) + (e & 3) + 8 * (e >> 2) + 4 * h;
        Es[lr * EP + 32 * ct + i] = acc[r][e] + bias_c;
      }
    }
    __syncthreads();
    const int nrows = min(128, TILE_ROWS - 128 * half);
    constexpr int RB = 8;                                 // rows in flight per wave: the residual loads are issued together
    for (int base = 0; base < nrows; base += 4 * RB) {
      float p[RB][VPL], gt[RB][VPL];
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int lr = base + 4 * j + w;
        const int64_t gr = row0 + 128 * half + lr;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
          p[j][k] = (a.f_prev && lr < nrows && gr < total_rows) ? a.f_prev[gr * COUT + lane * VPL + k] : 0.0f;
          // act 3 (svdd_conv1d_cl_gated_f32, the backward pass of a ReLU layer): a.tb is a ROW tensor [rows][COUT], the forward
          // activation whose sign gates this gradient
          gt[j][k] = (a.act == 3 && lr < nrows && gr < total_rows) ? a.tb[gr * COUT + lane * VPL + k] : 1.0f;
        }
      }
#pragma unroll
      for (int j = 0; j < RB; ++j) {
        const int lr = base + 4 * j + w;
        const int64_t gr = row0 + 128 * half + lr;
        if (lr >= nrows || gr >= total_rows) continue;    // wave-uniform
        const int64_t o = gr * COUT + lane * VPL;
        float v[VPL];
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < VPL; ++k) {
          const float t = Es[lr * EP + lane * VPL + k];
          v[k] = a.act == 0 ? fmaxf(t, 0.0f) + p[j][k] : a.act == 1 ? fmaxf(t + p[j][k], 0.0f) :
                 a.act == 3 ? (gt[j][k] > 0.0f ? t + p[j][k] : 0.0f) : t + p[j][k];
          a.y[o + k] = v[k];
          v[k] += tbv[k];
          sum += v[k];
        }
        if (a.hn) {
          const float mean = wave_sum(sum) * (1.0f / COUT);
          float q = 0.0f;
#pragma unroll
          for (int k = 0; k < VPL; ++k) { const float d = v[k] - mean; q += d * d; }
          const float rstd = rsqrtf(wave_sum(q) * (1.0f / COUT) + 1e-5f);
#pragma unroll
          for (int k = 0; k < VPL; ++k) a.hn[o + k] = (v[k] - mean) * rstd * gmv[k] + btv[k];
        }
      }
    }
  }
}

// ------------------------------------------------------------------- fused conv tower (value network) ----
// The whole ConvTower of the ConvGRU value net (reference Enformer.py:1634-1751 with the Stem :1754-1804 and five
// ConvBlocks "CDNRA" :2176-2292; eval-mode BatchNorm folded into the weights by the host):
//     a0 = relu(conv15(onehot) + b0) ;  a_{k+1} = relu(conv5(a_k) + b_k + a_k)   (k = 0..4, 64 channels)
// in ONE launch, one workgroup per tile of whole sequences (<= 208 rows). The activations never leave the CU:
// they ping-pong between two LDS images [rows][64] and are the MFMA A operands directly (a tap is a row offset;
// rows outside the sequence read a zero row), so there is no activation traffic to HBM between layers (5 x 2 x
// 131 MB per value forward at n = 2560), no separate bias/residual/ReLU pass and no im2col staging. Only the
// weights stream through LDS (one [64][32] tile per (layer, tap, 32-channel chunk), register-prefetched).
// v_mfma_f32_16x16x4_f32 with 16-row tiles: 13 row tiles x 4 column tiles = 52 units = 13 per SIMD exactly
// (200 rows pad to 208: 4 % waste instead of 12 % with 32-row tiles). 8 waves: wave w owns column tile w & 3 and
// the row tiles of parity w >> 2, so the two waves of a SIMD interleave their MFMA streams.
constexpr int TW_ROWS = 208;               // 13 row tiles of 16
constexpr int TW_RT = 13;
constexpr int TW_C = 64;
constexpr int TW_AP = TW_C + 4;            // LDS row stride of the activation images (floats)
constexpr int TW_MAXL = 8;                 // max conv layers after the stem

struct TowerArgs {
  const float* x;        // [n, L, 4] one-hot (value-function input)
  const float* tiles;    // [2 + 10*nlayers][64 cout][32] weight tiles in execution order:
                         //   stem: k = 4*tap + channel (60 real, 4 zero), chunks 0,1 ; then per layer, per 32-channel chunk, per tap
  const float* bias;     // [1 + nlayers][64]   stem bias, then the (BatchNorm-folded) layer biases
  float* out;            // [n, L, 64]
  int n, L, spt, nlayers, residual_mask;
  const int* count;      // device scalar: valid rows of a compacted batch (NULL: n) ; conv_tower_kernel only
};

// One LDS activation image only: a layer's outputs wait in the accumulators until every wave has finished reading
// the image, then each wave overwrites the positions it owns (reading its own residual first). 57 KB of LDS and
// <= 128 VGPRs per wave => TWO workgroups (16 waves) per CU, which overlap each other's barriers, LDS latency and
// global stores.
// Weight stream: each wave reads ITS [16 cout][32 k] slice of the (layer, chunk, tap) tile straight from L2 into the
// B-operand registers, one tile ahead: no LDS staging and no barrier per tap, so the waves of a workgroup run
// unsynchronised inside a layer. MFMA groups are fenced (sched_barrier) behind one explicit lgkmcnt wait: an
// s_waitcnt or ds_read between two MFMAs costs tens of cycles of matrix-pipe time on gfx950.
// SPT1 (one sequence per tile): image rows -1 and >= L are zero, so a tap is a clamped row offset.
template <bool SPT1>
__global__ __launch_bounds__(512, 4) void conv_tower_kernel(TowerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem + TW_AP;                            // rows -1 .. TW_ROWS ; [-1] and [TW_ROWS] stay zero
  float* xs = smem + (TW_ROWS + 2) * TW_AP + 4;         // one-hot tile, rows -1 .. TW_ROWS ; [-1], [>= L], [TW_ROWS] zero

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cs = w & 3, rh = w >> 2;                    // column tile (16 channels) ; row-tile parity
  const int j = lane & 15, g = lane >> 4;
  const int L = a.L;
  const int tile_rows = a.spt * L;
  const int64_t row0 = (int64_t)blockIdx.x * tile_rows;
  const int64_t total_rows = (int64_t)(a.count ? __builtin_amdgcn_readfirstlane(*a.count) : a.n) * L;
  if (row0 >= total_rows) return;

  for (int e = tid - 1; e < TW_ROWS + 1; e += 512) {
    float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (e >= 0 && e < tile_rows && row0 + e < total_rows) v = *reinterpret_cast<const float4*>(a.x + (row0 + e) * 4);
    *reinterpret_cast<float4*>(xs + 4 * e) = v;
  }
  for (int e = tid; e < TW_AP; e += 512) { smem[e] = 0.0f; act[TW_ROWS * TW_AP + e] = 0.0f; }
  const float* wsrc = a.tiles + (16 * cs + j) * CH + 8 * g;
  float4 bn0 = *reinterpret_cast<const float4*>(wsrc), bn1 = *reinterpret_cast<const float4*>(wsrc + 4);

  const int arow0 = 16 * rh + j;                        // this lane feeds rows arow0 + 32 r as the A operand
  int apos[SPT1 ? 1 : 7];
  if (!SPT1) {
#pragma unroll
    for (int r = 0; r < 7; ++r)
      apos[r] = (rh + 2 * r < TW_RT && arow0 + 32 * r < tile_rows) ? (arow0 + 32 * r) % L : -(1 << 20);
  }
  const int abase = ((16 * rh + j) * TW_AP + 8 * g) * 4;             // byte offset of (row 16 rh + j, col 8 g)
  const int a_lo = abase - (16 * rh + j + 1) * TW_AP * 4;            // row -1
  const int a_hi = abase + (TW_ROWS - 16 * rh - j) * TW_AP * 4;      // row TW_ROWS
  const char* actb = reinterpret_cast<const char*>(act);
  const int nit = 2 + 10 * a.nlayers;
  int it = 0;
  f32x4 acc[7];
  __syncthreads();                                       // xs and the zero rows are visible

#define TW_WAIT(NOUT) __builtin_amdgcn_s_waitcnt(0xC07F | ((NOUT) << 8));
#define TW_MMA(AF, R0, R1)                                                                                    \
  _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                             \
    _Pragma("unroll") for (int r = R0; r < R1; ++r) if (!(r == 6 && rh == 1)) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r - R0][q].x, bf[4 * q], acc[r], 0, 0, 0);     \
    _Pragma("unroll") for (int r = R0; r < R1; ++r) if (!(r == 6 && rh == 1)) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r - R0][q].y, bf[4 * q + 1], acc[r], 0, 0, 0); \
    _Pragma("unroll") for (int r = R0; r < R1; ++r) if (!(r == 6 && rh == 1)) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r - R0][q].z, bf[4 * q + 2], acc[r], 0, 0, 0); \
    _Pragma("unroll") for (int r = R0; r < R1; ++r) if (!(r == 6 && rh == 1)) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r - R0][q].w, bf[4 * q + 3], acc[r], 0, 0, 0); \
  }

  for (int layer = -1; layer < a.nlayers; ++layer) {     // layer -1 = the stem (15 taps x 4 channels as K = 64)
    const float bl = a.bias[(layer + 1) * TW_C + 16 * cs + j];
#pragma unroll
    for (int r = 0; r < 7; ++r) acc[r] = f32x4{bl, bl, bl, bl};
    const int niter = layer < 0 ? 2 : 10;
    for (int ci = 0; ci < niter; ++ci, ++it) {
      const float bf[8] = {bn0.x, bn0.y, bn0.z, bn0.w, bn1.x, bn1.y, bn1.z, bn1.w};
      if (it + 1 < nit) {                                // the next tile's slice flies under the MFMAs
        const float* src = wsrc + (size_t)(it + 1) * TW_C * CH;
        bn0 = *reinterpret_cast<const float4*>(src);
        bn1 = *reinterpret_cast<const float4*>(src + 4);
      }
      if (layer < 0) {
        // k = 32 ci + 8 g + s covers taps t0 = 8 ci + 2 g and t0 + 1 (4 channels each): two float4 of the one-hot tile
        const int t0 = 8 * ci + 2 * g;
        float4 af[4][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int r0 = half ? 4 : 0, r1 = half ? 7 : 4;
#pragma unroll
          for (int r = r0; r < r1; ++r) {
            if (r == 6 && rh == 1) continue;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int rr = arow0 + 32 * r + t0 + q - 7;
              int idx;
              if (SPT1) idx = min(max(rr, -1), TW_ROWS);
              else idx = (unsigned)(apos[SPT1 ? 0 : r] + t0 + q - 7) < (unsigned)L ? rr : TW_ROWS;
              af[r - r0][q] = *reinterpret_cast<const float4*>(xs + 4 * idx);
            }
          }
          if (half == 0) { TW_MMA(af, 0, 4) } else { TW_MMA(af, 4, 7) }
        }
      } else {
        const int c = ci / 5, delta = ci - 5 * c - 2;
        const int dbytes = delta * (TW_AP * 4) + c * (CH * 4);
        // tile groups (0,1,2)(3,4)(5,6) through two register buffers, each group's reads one group ahead
        float4 fa[3][2], fb[2][2];
#define TW_ALOAD(R, V)                                                                                        \
        if (!((R) == 6 && rh == 1)) {                                                                         \
          int o_;                                                                                             \
          if (SPT1) o_ = min(max(abase + dbytes + (R) * (32 * TW_AP * 4), a_lo + c * (CH * 4)), a_hi + c * (CH * 4)); \
          else o_ = (unsigned)(apos[SPT1 ? 0 : (R)] + delta) < (unsigned)L ? abase + dbytes + (R) * (32 * TW_AP * 4) \
                                                                         : a_hi + c * (CH * 4);               \
          const float4* ap_ = reinterpret_cast<const float4*>(actb + o_);                                     \
          V[0] = ap_[0]; V[1] = ap_[1];                                                                       \
        }
        TW_ALOAD(0, fa[0]) TW_ALOAD(1, fa[1]) TW_ALOAD(2, fa[2])
        TW_ALOAD(3, fb[0]) TW_ALOAD(4, fb[1])
        __builtin_amdgcn_sched_barrier(0);
        TW_WAIT(4)
        TW_MMA(fa, 0, 3)
        __builtin_amdgcn_sched_barrier(0);
        TW_ALOAD(5, fa[0]) TW_ALOAD(6, fa[1])
        __builtin_amdgcn_sched_barrier(0);
        if (rh == 1) { TW_WAIT(2) } else { TW_WAIT(4) }
        TW_MMA(fb, 3, 5)
        __builtin_amdgcn_sched_barrier(0);
        TW_WAIT(0)
        TW_MMA(fa, 5, 7)
        __builtin_amdgcn_sched_barrier(0);
#undef TW_ALOAD
      }
    }
    // every wave must be done reading the image before its owners overwrite it (the stem reads xs, not the image)
    if (layer >= 0) __syncthreads();
    const bool res = layer >= 0 && ((a.residual_mask >> layer) & 1);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      if (r == 6 && rh == 1) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {                      // C/D layout: reg e -> row 4 g + e, column j
        const int row = 16 * (rh + 2 * r) + 4 * g + e;
        const int o = row * TW_AP + 16 * cs + j;
        const float v = acc[r][e] + (res ? act[o] : 0.0f);
        act[o] = row < tile_rows ? fmaxf(v, 0.0f) : 0.0f;
      }
    }
    __syncthreads();                                     // the image is complete
  }
#undef TW_MMA
#undef TW_WAIT
  for (int e = tid; e < tile_rows * 16; e += 512) {      // final image -> HBM, 16 B per thread, rows contiguous
    const int row = e >> 4, q = e & 15;
    if (row0 + row < total_rows)
      *reinterpret_cast<float4*>(a.out + (row0 + row) * TW_C + 4 * q) = *reinterpret_cast<const float4*>(act + row * TW_AP + 4 * q);
  }
}

// ---------------------------------------------- conv tower on candidate WINDOWS (SVDD-MC: M candidates per parent) ----
// The M candidates of a sample differ from their parent x_t only where a MASK was replaced this step (~L/steps
// positions), and the tower's receptive field is +-17 rows (stem 15 taps + 5 x 5 taps). So tower(candidate) equals
// tower(parent) outside +-17 rows of the changed positions, bit for bit (every output element accumulates the same
// products in the same order). This kernel computes, per candidate, only a contiguous row window [w0, w1) (multiples
// of 16) that covers the changed positions +-27 rows: +-17 for the receptive field, +10 because rows next to a window
// edge see zeros instead of their real neighbours and are wrong by up to 2 rows per conv layer (the stem reads the real
// one-hot rows beyond the window). Rows [w0 + 10, w1 - 10) come from this computation (a window edge that is a
// sequence end is exact), all other rows are copied from the parent's tower output (computed once per parent by
// conv_tower_kernel). At L = 200, 128 steps, M = 10: 42 % of the row tiles, 21 % of the candidates need none at all.
struct TowerWinArgs {
  TowerArgs t;             // x = candidates' one-hot [n, L, 4], out [n, L, 64]
  const int* win;          // [n][2]  (w0, w1) ; w0 == w1: the candidate equals its parent
  const float* parent_out; // [n / M, L, 64]
  int M;
  const int* live_idx;     // [count] candidate ids to process (NULL: identity); workgroup i writes out[i]
  const int* count;        // device scalar (NULL: n)
};

__global__ __launch_bounds__(512, 4) void conv_tower_win_kernel(TowerWinArgs wa) {
  const TowerArgs& a = wa.t;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem + TW_AP;                            // rows -1 .. TW_ROWS ; [-1] and [TW_ROWS] stay zero
  float* xs = smem + (TW_ROWS + 2) * TW_AP + 8 * 4;     // one-hot rows -8 .. TW_ROWS + 8 (real data beyond the window)

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cs = w & 3, rh = w >> 2;
  const int j = lane & 15, g = lane >> 4;
  const int L = a.L;
  if (wa.count && (int)blockIdx.x >= __builtin_amdgcn_readfirstlane(*wa.count)) return;
  const int cand = wa.live_idx ? __builtin_amdgcn_readfirstlane(wa.live_idx[blockIdx.x]) : (int)blockIdx.x;
  const int w0 = __builtin_amdgcn_readfirstlane(wa.win[2 * cand]);
  const int w1 = __builtin_amdgcn_readfirstlane(wa.win[2 * cand + 1]);
  const int nt = (w1 - w0) >> 4;                        // live row tiles
  const int lrows = min(L, w1) - w0;                    // valid local rows
  const int keep_lo = nt == 0 ? 0 : (w0 == 0 ? 0 : w0 + 10);
  const int keep_hi = nt == 0 ? 0 : (w1 >= L ? L : w1 - 10);
  float* outc = a.out + (size_t)blockIdx.x * L * TW_C;
  const float* par = wa.parent_out + (size_t)(cand / wa.M) * L * TW_C;

  for (int e = tid; e < L * 16; e += 512) {             // rows that are the parent's, straight from its output
    const int row = e >> 4;
    if (row < keep_lo || row >= keep_hi)
      *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * (e & 15)) = *reinterpret_cast<const float4*>(par + (size_t)row * TW_C + 4 * (e & 15));
  }
  if (nt == 0) return;

  const float* xc = a.x + (size_t)cand * L * 4;
  for (int e = tid - 8; e < TW_ROWS + 8; e += 512) {
    const int gl = w0 + e;
    float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if (gl >= 0 && gl < L) v = *reinterpret_cast<const float4*>(xc + (size_t)gl * 4);
    *reinterpret_cast<float4*>(xs + 4 * e) = v;
  }
  for (int e = tid; e < (TW_ROWS + 2) * TW_AP; e += 512) smem[e] = 0.0f;     // rows beyond the window must read as zero
  const float* wsrc = a.tiles + (16 * cs + j) * CH + 8 * g;
  float4 bn0 = *reinterpret_cast<const float4*>(wsrc), bn1 = *reinterpret_cast<const float4*>(wsrc + 4);

  const int arow0 = 16 * rh + j;
  const int abase = ((16 * rh + j) * TW_AP + 8 * g) * 4;
  const int a_lo = abase - (16 * rh + j + 1) * TW_AP * 4;
  const int a_hi = abase + (TW_ROWS - 16 * rh - j) * TW_AP * 4;
  const char* actb = reinterpret_cast<const char*>(act);
  const int nlive = (nt - rh + 1) >> 1;                 // owned live tiles rh + 2 r, r < nlive (a prefix)
  const int nit = 2 + 10 * a.nlayers;
  int it = 0;
  f32x4 acc[7];
  __syncthreads();

#define TW_WAIT(NOUT) __builtin_amdgcn_s_waitcnt(0xC07F | ((NOUT) << 8));
#define TWW_MMA(AF, R0, CNT)                                                                                  \
  _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                             \
    _Pragma("unroll") for (int r = 0; r < CNT; ++r) acc[R0 + r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r][q].x, bf[4 * q], acc[R0 + r], 0, 0, 0);     \
    _Pragma("unroll") for (int r = 0; r < CNT; ++r) acc[R0 + r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r][q].y, bf[4 * q + 1], acc[R0 + r], 0, 0, 0); \
    _Pragma("unroll") for (int r = 0; r < CNT; ++r) acc[R0 + r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r][q].z, bf[4 * q + 2], acc[R0 + r], 0, 0, 0); \
    _Pragma("unroll") for (int r = 0; r < CNT; ++r) acc[R0 + r] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[r][q].w, bf[4 * q + 3], acc[R0 + r], 0, 0, 0); \
  }
#define TWW_GROUP3(AF, R0) { const int c_ = nlive - (R0); if (c_ >= 3) { TWW_MMA(AF, R0, 3) } else if (c_ == 2) { TWW_MMA(AF, R0, 2) } else if (c_ == 1) { TWW_MMA(AF, R0, 1) } }
#define TWW_GROUP2(AF, R0) { const int c_ = nlive - (R0); if (c_ >= 2) { TWW_MMA(AF, R0, 2) } else if (c_ == 1) { TWW_MMA(AF, R0, 1) } }

  for (int layer = -1; layer < a.nlayers; ++layer) {
    const float bl = a.bias[(layer + 1) * TW_C + 16 * cs + j];
#pragma unroll
    for (int r = 0; r < 7; ++r) acc[r] = f32x4{bl, bl, bl, bl};
    const int niter = layer < 0 ? 2 : 10;
    for (int ci = 0; ci < niter; ++ci, ++it) {
      const float bf[8] = {bn0.x, bn0.y, bn0.z, bn0.w, bn1.x, bn1.y, bn1.z, bn1.w};
      if (it + 1 < nit) {
        const float* src = wsrc + (size_t)(it + 1) * TW_C * CH;
        bn0 = *reinterpret_cast<const float4*>(src);
        bn1 = *reinterpret_cast<const float4*>(src + 4);
      }
      if (layer < 0) {
        const int t0 = 8 * ci + 2 * g;
        float4 fa[3][2], fb[2][2];
#define TWW_XLOAD(R, V)                                                                                       \
        { const float* xp_ = xs + 4 * (arow0 + 32 * (R) + t0 - 7);                                            \
          V[0] = *reinterpret_cast<const float4*>(xp_); V[1] = *reinterpret_cast<const float4*>(xp_ + 4); }
        TWW_XLOAD(0, fa[0]) TWW_XLOAD(1, fa[1]) TWW_XLOAD(2, fa[2])
        TWW_XLOAD(3, fb[0]) TWW_XLOAD(4, fb[1])
        __builtin_amdgcn_sched_barrier(0);
        TW_WAIT(4)
        TWW_GROUP3(fa, 0)
        __builtin_amdgcn_sched_barrier(0);
        TWW_XLOAD(5, fa[0])
        if (rh == 0) TWW_XLOAD(6, fa[1])
        __builtin_amdgcn_sched_barrier(0);
        if (rh == 1) { TW_WAIT(2) } else { TW_WAIT(4) }
        TWW_GROUP2(fb, 3)
        __builtin_amdgcn_sched_barrier(0);
        TW_WAIT(0)
        TWW_GROUP2(fa, 5)
        __builtin_amdgcn_sched_barrier(0);
#undef TWW_XLOAD
      } else {
        const int c = ci / 5, delta = ci - 5 * c - 2;
        const int dbytes = delta * (TW_AP * 4) + c * (CH * 4);
        float4 fa[3][2], fb[2][2];
#define TWW_ALOAD(R, V)                                                                                       \
        { const int o_ = min(max(abase + dbytes + (R) * (32 * TW_AP * 4), a_lo + c * (CH * 4)), a_hi + c * (CH * 4)); \
          const float4* ap_ = reinterpret_cast<const float4*>(actb + o_);                                     \
          V[0] = ap_[0]; V[1] = ap_[1]; }
        TWW_ALOAD(0, fa[0]) TWW_ALOAD(1, fa[1]) TWW_ALOAD(2, fa[2])
        TWW_ALOAD(3, fb[0]) TWW_ALOAD(4, fb[1])
        __builtin_amdgcn_sched_barrier(0);
        TW_WAIT(4)
        TWW_GROUP3(fa, 0)
        __builtin_amdgcn_sched_barrier(0);
        TWW_ALOAD(5, fa[0])
        if (rh == 0) TWW_ALOAD(6, fa[1])
        __builtin_amdgcn_sched_barrier(0);
        if (rh == 1) { TW_WAIT(2) } else { TW_WAIT(4) }
        TWW_GROUP2(fb, 3)
        __builtin_amdgcn_sched_barrier(0);
        TW_WAIT(0)
        TWW_GROUP2(fa, 5)
        __builtin_amdgcn_sched_barrier(0);
#undef TWW_ALOAD
      }
    }
    if (layer >= 0) __syncthreads();
    const bool res = layer >= 0 && ((a.residual_mask >> layer) & 1);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      if (r >= nlive) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * (rh + 2 * r) + 4 * g + e;
        const int o = row * TW_AP + 16 * cs + j;
        const float v = acc[r][e] + (res ? act[o] : 0.0f);
        act[o] = row < lrows ? fmaxf(v, 0.0f) : 0.0f;
      }
    }
    __syncthreads();
  }
#undef TWW_GROUP2
#undef TWW_GROUP3
#undef TWW_MMA
#undef TW_WAIT
  for (int e = tid; e < L * 16; e += 512) {
    const int row = e >> 4, q = e & 15;
    if (row >= keep_lo && row < keep_hi)
      *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * q) = *reinterpret_cast<const float4*>(act + (row - w0) * TW_AP + 4 * q);
  }
}

// ----------------------------------------------------- conv tower, second generation (whole sequences AND windows) ----
// Same function, same bits as conv_tower_kernel / conv_tower_win_kernel (every output element accumulates the same
// products in the same order), restructured after what the 16-bit twin taught (svdd_lp_tower.hip):
//   * wave w owns 32 output channels (column tiles 2 cp, 2 cp + 1, cp = w & 1) of the row tiles rq + 4 r (rq = w >> 1), so
//     an A fragment (two ds_read_b128) feeds 16 MFMAs instead of 8: half the LDS read traffic;
//   * the number of live row tiles of a wave (4 / 3 for whole sequences, 0..4 for a window) is a TEMPLATE parameter of the
//     layer loop, chosen once per wave: straight-line code, no per-tile predicates;
//   * <= 128 VGPRs: two workgroups per CU.
struct TowerCtx2 {
  float* act;            // LDS image, row 0 ; rows -1 and TW_ROWS are zero
  const float* xs;       // LDS one-hot rows [row][4]
  const float* wsrc;     // this lane's slice of tile 0
  const float* bias;
  int L, tile_rows, nlayers, residual_mask;
  int rq, cp, j, g;
  int apos[7];
};

// CT = column tiles (16 output channels each) per wave: the 8 waves are 4 / CT channel groups x 2 CT row groups, wave
// (rq, cp) computes row tiles rq + 2 CT r, r < NL.
template <bool CLAMP, int NL, int CT>
__device__ __forceinline__ void tower2_layers(const TowerCtx2& c) {
  constexpr int RS = 2 * CT;                                     // row-tile stride of a wave
  constexpr int NA = NL > 0 ? NL : 1;
  const int j = c.j, g = c.g, L = c.L;
  const int arow0 = 16 * c.rq + j;
  const int abase = (arow0 * TW_AP + 8 * g) * 4;                 // byte offset of (row arow0, channel 8 g)
  const int a_lo = abase - (arow0 + 1) * TW_AP * 4;              // row -1
  const int a_hi = abase + (TW_ROWS - arow0) * TW_AP * 4;        // row TW_ROWS
  const char* actb = reinterpret_cast<const char*>(c.act);
  const int cb = 16 * CT * c.cp + 4 * g;                         // this lane's OUTPUT channels: cb + 16 ct + e (transposed tiles)
  const int nit = 2 + 10 * c.nlayers;
  float4 bn[2 * CT];                                             // [ct][half]: W[ch0 + 16 ct][8 g .. 8 g + 8] of the next tile
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    bn[2 * ct] = *reinterpret_cast<const float4*>(c.wsrc + ct * 16 * CH);
    bn[2 * ct + 1] = *reinterpret_cast<const float4*>(c.wsrc + ct * 16 * CH + 4);
  }
  int it = 0;
  f32x4 acc[NA][CT];

  for (int layer = -1; layer < c.nlayers; ++layer) {
    // TRANSPOSED accumulators (round 4d, as in the 16-bit twin): the weight fragment is the A operand, the activation fragment
    // the B operand — the same registers either way and the same products in the same order — so lane (j, g) register e of
    // column tile ct holds channel cb + 16 ct + e, cb = 16 CT cp + 4 g, of ONE position (row 16 (rq + RS r) + j): the
    // epilogue is one 16-byte LDS load and store per (row tile, column tile) instead of four 4-byte ones
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const f32x4 bl = *reinterpret_cast<const f32x4*>(c.bias + (layer + 1) * TW_C + cb + 16 * ct);
#pragma unroll
      for (int r = 0; r < NL; ++r) acc[r][ct] = bl;
    }
    const int niter = layer < 0 ? 2 : 10;
    for (int ci = 0; ci < niter; ++ci, ++it) {
      float b[CT][8];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        b[ct][0] = bn[2 * ct].x; b[ct][1] = bn[2 * ct].y; b[ct][2] = bn[2 * ct].z; b[ct][3] = bn[2 * ct].w;
        b[ct][4] = bn[2 * ct + 1].x; b[ct][5] = bn[2 * ct + 1].y; b[ct][6] = bn[2 * ct + 1].z; b[ct][7] = bn[2 * ct + 1].w;
      }
      if (it + 1 < nit) {
        const float* src = c.wsrc + (size_t)(it + 1) * TW_C * CH;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          bn[2 * ct] = *reinterpret_cast<const float4*>(src + ct * 16 * CH);
          bn[2 * ct + 1] = *reinterpret_cast<const float4*>(src + ct * 16 * CH + 4);
        }
      }
      float4 af[NA][2];
      if (layer < 0) {
        const int t0 = 8 * ci + 2 * g;                           // k = 32 ci + 8 g + s: taps t0, t0 + 1 (4 channels each)
#pragma unroll
        for (int r = 0; r < NL; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int rr = arow0 + 16 * RS * r + t0 + q - 7;
            int idx;
            if (CLAMP) idx = rr;                                 // xs holds zero rows outside the sequence / window
            else idx = (unsigned)(c.apos[r] + t0 + q - 7) < (unsigned)L ? rr : TW_ROWS + 7;
            af[r][q] = *reinterpret_cast<const float4*>(c.xs + 4 * idx);
          }
      } else {
        const int chk = ci / 5, delta = ci - 5 * chk - 2;
        const int dbytes = delta * (TW_AP * 4) + chk * (CH * 4);
#pragma unroll
        for (int r = 0; r < NL; ++r) {
          int o;
          if (CLAMP) o = min(max(abase + dbytes + r * (16 * RS * TW_AP * 4), a_lo + chk * (CH * 4)), a_hi + chk * (CH * 4));
          else o = (unsigned)(c.apos[r] + delta) < (unsigned)L ? abase + dbytes + r * (16 * RS * TW_AP * 4) : a_hi + chk * (CH * 4);
          const float4* ap = reinterpret_cast<const float4*>(actb + o);
          af[r][0] = ap[0]; af[r][1] = ap[1];
        }
      }
#pragma unroll
      for (int r = 0; r < NL; ++r)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float av[4] = {af[r][q].x, af[r][q].y, af[r][q].z, af[r][q].w};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
              acc[r][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ct][4 * q + s4], av[s4], acc[r][ct], 0, 0, 0);
        }
    }
    if (layer >= 0) __syncthreads();                     // every wave is done reading the image (the stem reads xs)
    const bool res = layer >= 0 && ((c.residual_mask >> layer) & 1);
#pragma unroll
    for (int r = 0; r < NL; ++r) {
      const int row = 16 * (c.rq + RS * r) + j;          // C/D layout, transposed: reg e -> channel cb + 16 ct + e at position j
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        f32x4* dst = reinterpret_cast<f32x4*>(c.act + row * TW_AP + cb + 16 * ct);
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4 v = acc[r][ct] + (res ? *dst : z);
        *dst = row < c.tile_rows ? __builtin_elementwise_max(v, z) : z;
      }
    }
    __syncthreads();                                     // the image is complete
  }
}

template <bool CLAMP, bool WIN, int CT>
__device__ __forceinline__ void tower2_dispatch(const TowerCtx2& c, int nlive) {
  if constexpr (CT == 2) {
    switch (nlive) {                                    // wave-uniform; every path runs the same barriers
      case 4: tower2_layers<CLAMP, 4, 2>(c); break;
      case 3: tower2_layers<CLAMP, 3, 2>(c); break;
      case 2: if constexpr (WIN) tower2_layers<CLAMP, 2, 2>(c); break;
      case 1: if constexpr (WIN) tower2_layers<CLAMP, 1, 2>(c); break;
      default: if constexpr (WIN) tower2_layers<CLAMP, 0, 2>(c); break;
    }
  } else {
    switch (nlive) {
      case 7: tower2_layers<CLAMP, 7, 1>(c); break;
      case 6: tower2_layers<CLAMP, 6, 1>(c); break;
      case 5: if constexpr (WIN) tower2_layers<CLAMP, 5, 1>(c); break;
      case 4: if constexpr (WIN) tower2_layers<CLAMP, 4, 1>(c); break;
      case 3: if constexpr (WIN) tower2_layers<CLAMP, 3, 1>(c); break;
      case 2: if constexpr (WIN) tower2_layers<CLAMP, 2, 1>(c); break;
      case 1: if constexpr (WIN) tower2_layers<CLAMP, 1, 1>(c); break;
      default: if constexpr (WIN) tower2_layers<CLAMP, 0, 1>(c); break;
    }
  }
}

template <bool SPT1, bool WIN, int CT>
__global__ __launch_bounds__(512, 4) void conv_tower2_kernel(TowerWinArgs wa) {
  const TowerArgs& a = wa.t;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* act = smem + TW_AP;                            // rows -1 .. TW_ROWS ; [-1] and [TW_ROWS] stay zero
  float* xs = smem + (TW_ROWS + 2) * TW_AP + 8 * 4;     // one-hot rows -8 .. TW_ROWS + 8

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.L;
  int cand = blockIdx.x, w0 = 0, w1 = 0;
  if (WIN) {
    if (wa.count && (int)blockIdx.x >= __builtin_amdgcn_readfirstlane(*wa.count)) return;
    if (wa.live_idx) cand = __builtin_amdgcn_readfirstlane(wa.live_idx[blockIdx.x]);
    w0 = __builtin_amdgcn_readfirstlane(wa.win[2 * cand]);
    w1 = __builtin_amdgcn_readfirstlane(wa.win[2 * cand + 1]);
  }
  const int nt = (w1 - w0) >> 4;
  const int tile_rows = WIN ? min(L, w1) - w0 : a.spt * L;
  const int64_t row0 = WIN ? 0 : (int64_t)blockIdx.x * tile_rows;
  const int64_t total_rows = (int64_t)((!WIN && a.count) ? __builtin_amdgcn_readfirstlane(*a.count) : a.n) * L;
  if (!WIN && row0 >= total_rows) return;
  const int keep_lo = !WIN ? 0 : (nt == 0 ? 0 : (w0 == 0 ? 0 : w0 + 10));
  const int keep_hi = !WIN ? 0 : (nt == 0 ? 0 : (w1 >= L ? L : w1 - 10));
  float* outc = a.out + (WIN ? (size_t)blockIdx.x * L * TW_C : 0);

  if (WIN) {
    const float* par = wa.parent_out + (size_t)(cand / wa.M) * L * TW_C;
    for (int e = tid; e < L * 16; e += 512) {           // rows that are the parent's, straight from its output
      const int row = e >> 4;
      if (row < keep_lo || row >= keep_hi)
        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * (e & 15)) = *reinterpret_cast<const float4*>(par + (size_t)row * TW_C + 4 * (e & 15));
    }
    if (nt == 0) return;
  }
  {
    const float* xc = a.x + (WIN ? (size_t)cand * L * 4 : 0);
    for (int e = tid - 8; e < TW_ROWS + 8; e += 512) {
      float4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (WIN) { const int gl = w0 + e; if (gl >= 0 && gl < L) v = *reinterpret_cast<const float4*>(xc + (size_t)gl * 4); }
      else if (e >= 0 && e < tile_rows && row0 + e < total_rows) v = *reinterpret_cast<const float4*>(xc + (row0 + e) * 4);
      *reinterpret_cast<float4*>(xs + 4 * e) = v;
    }
  }
  for (int e = tid; e < (TW_ROWS + 2) * TW_AP; e += 512) smem[e] = 0.0f;      // image incl. the zero rows

  TowerCtx2 c;
  c.act = act; c.xs = xs; c.bias = a.bias;
  c.L = L; c.tile_rows = tile_rows; c.nlayers = a.nlayers; c.residual_mask = a.residual_mask;
  constexpr int CG = 4 / CT, RS = 2 * CT;               // channel groups ; row-tile stride
  c.cp = w % CG; c.rq = w / CG; c.j = lane & 15; c.g = lane >> 4;
  c.wsrc = a.tiles + (16 * CT * c.cp + c.j) * CH + 8 * c.g;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int row = 16 * (c.rq + RS * r) + c.j;
    c.apos[r] = (!SPT1 && !WIN && row < tile_rows) ? row % L : -(1 << 20);
  }
  const int nlive = ((WIN ? nt : TW_ROWS / 16) - c.rq + RS - 1) / RS;
  __syncthreads();
  tower2_dispatch<SPT1 || WIN, WIN, CT>(c, nlive);
  if (WIN) {
    for (int e = tid; e < L * 16; e += 512) {
      const int row = e >> 4, q = e & 15;
      if (row >= keep_lo && row < keep_hi)
        *reinterpret_cast<float4*>(outc + (size_t)row * TW_C + 4 * q) = *reinterpret_cast<const float4*>(act + (row - w0) * TW_AP + 4 * q);
    }
  } else {
    for (int e = tid; e < tile_rows * 16; e += 512) {
      const int row = e >> 4, q = e & 15;
      if (row0 + row < total_rows)
        *reinterpret_cast<float4*>(a.out + (row0 + row) * TW_C + 4 * q) = *reinterpret_cast<const float4*>(act + row * TW_AP + 4 * q);
    }
  }
}

// (w0, w1) of every candidate: one wave per candidate compares it with its parent.
__global__ __launch_bounds__(256) void candidate_windows_kernel(const uint8_t* __restrict__ cand, const uint8_t* __restrict__ x,
                                                               int n, int L, int M, int margin, int* __restrict__ win,
                                                               int* __restrict__ flags) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n) return;
  const int lane = threadIdx.x & 63;
  const uint8_t* cp = cand + (size_t)c * L;
  const uint8_t* xp = x + (size_t)(c / M) * L;
  int lo = 1 << 30, hi = -1;
  for (int l = lane; l < L; l += 64)
    if (cp[l] != xp[l]) { lo = min(lo, l); hi = max(hi, l); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off, 64)); hi = max(hi, __shfl_xor(hi, off, 64)); }
  if (lane == 0) {
    int w0 = 0, w1 = 0;
    if (hi >= 0) {
      w0 = max(0, lo - margin) & ~15;
      w1 = min((L + 15) & ~15, (hi + margin + 1 + 15) & ~15);
    }
    win[2 * c] = w0; win[2 * c + 1] = w1;
    if (flags) flags[c] = (w1 - w0) >> 4;              // row tiles of the window (>= 1) ; 0: the candidate is a copy of its parent
  }
}

// --------------------------------------------------------- fused dilated-CNN backbone (one launch per forward) ----
// The whole masked-diffusion backbone of the reference (models/dnaconv.py:176-210, sigma = 0):
//     f_0 = relu(conv9(onehot5(x)) + b)                      first layer as a 9-entry table lookup per output
//     hn_i = LayerNorm(f_i + tb_i) ; f_{i+1} = relu(conv9_dil_i(hn_i) + b_i) + f_i          i = 0..nl-1
//     logits = W2 relu(W1 f_nl + b1) + b2
// One workgroup per tile of whole sequences (<= 208 rows), 8 waves; every wave keeps its part of the residual stream f
// in registers, in the MFMA C/D layout, for the whole forward (work split: see the kernel). The only LDS-resident
// activation is the LayerNorm'd image hn [rows][128] that feeds the MFMA A operands (a tap = a row offset;
// out-of-sequence rows read a zero row). No activation ever goes to HBM: per forward the layer-wise path moved ~3 GB
// (conv in/out + the epilogue/LayerNorm pass per layer) and launched 41 kernels.
// v_mfma_f32_16x16x4_f32, 16-row tiles (200 rows -> 13 tiles, 4 % padding); row tiles whose rows all fall into the
// zero padding of a dilated tap are skipped (runtime, workgroup-uniform, from a schedule built once in LDS).
// Weights are NOT staged through LDS: each wave reads its [32 cout][32 k] slice of the (layer, chunk, tap) tile
// straight from L2 into the B-operand registers, one tile ahead, so there is no barrier inside a layer.
// LayerNorm statistics are two-pass (mean, then centred variance) with a 16-lane DPP reduction and a 4-way LDS
// exchange. Measurements and the PMC-guided history of this kernel: DESIGN.md section 4.
constexpr int BB_C = 128;
constexpr int BB_AP = BB_C + 4;
constexpr int BB_MAXL = 32;

struct BackboneArgs {
  const uint8_t* x;        // [n, L] tokens 0..4
  const float* table0;     // [9][5][128]  W_first[co][c][t] -> [t][c][co]
  const float* tiles;      // [nl][4 chunks][9 taps][128][32] then final-1x1 [4][128][32]
  const float* vec;        // [nl + 2][4][128] : row 0: {b_first,-,-,-}; row 1+i: {bias_i, tb_i, gamma_i, beta_i}; last: {b_f1,-,-,-}
  const float* w2;         // [5][128] , then b2 [5]
  float* out;              // [n, L, 5]
  int n, L, spt, nl;
  int dil[BB_MAXL];
  const int* count;        // device scalar: valid rows (NULL: n) — exact work-skipping on a compacted batch
  const int* row_idx;      // [count] (NULL: identity): compact row r reads the tokens of sequence row_idx[r] of x ...
  int out_scatter;         // ... and writes its logits to row row_idx[r] of out (1) or to row r (0)
  int auto_spt, ncu;       // auto_spt: the workgroups pick the sequences per tile from the device-side row count (svdd_spt.h)
  SvddTilePlan plan;       // several sequences per tile, row count known on the host: which tile takes how many (svdd_spt.h)
};

// What the gradient kernel (backbone_grad_kernel, DPS) needs from a forward, all in the forward kernel's LANE-PRIVATE layout — lane
// `tid` of workgroup `wg` owns element q = (r * 2 + ct) * 4 + e, q < 56: row 16 (rh + 2 r) + 4 g + e, column 32 cg + j + 16 ct —
// so every store / load of the pair of kernels is one fully coalesced 256-byte access per wave:
//   xhat [n][nl][56][512]  the LayerNorm'd value before the affine map, (h - mean) rstd, of every layer's input
//   rstd [n][nl][208]      1 / sqrt(var + eps) per row
//   mask [n][nl + 2][512]  bit q of a lane's 64-bit word: ReLU decision of the first layer (word 0), of layer i's conv (1 + i), of
//                          final_conv's first 1x1 (nl + 1)
struct BackboneSave {
  float* xhat;
  float* rstd;
  unsigned long long* mask;
};

template <int N>
__device__ __forceinline__ float row_ror(float v) {            // rotate right by N inside each 16-lane DPP row
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float group16_sum(float v) {        // sum over the 16 lanes sharing lane >> 4 (VALU DPP, no LDS)
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}

// Work split: wave w owns 32 output channels (column group w & 3) of the row tiles of parity w >> 2 (7 / 6 tiles; the
// two waves of a SIMD share a column group and split the rows). An MFMA group is ONE row tile x two column tiles x
// 8 k-steps = 16 MFMAs alternating two accumulators, so
//   * an A fragment (two ds_read_b128 + one clamped address) feeds two column tiles;
//   * dead row tiles are skipped one by one;
//   * a LayerNorm row has two of its values in the same lane (half the DPP reductions, 4 partials per row).
// (A first version gave each wave 16 channels of all 13 row tiles and skipped tiles in pairs: twice the LDS reads and
//  address VALU per MFMA, 6 % more MFMAs; 2.31 vs 2.17 ms.)
// LDS image rows: row -1 and rows >= L of a one-sequence tile are zero, so a tap is a clamped row offset.
template <bool SPT1, bool SAVE = false>
__global__ __launch_bounds__(512, 2) void backbone_kernel(BackboneArgs a, BackboneSave sv) {
  static_assert(SPT1 || !SAVE, "the saving forward is the one-sequence-per-tile form");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img = smem + BB_AP;                              // rows -1 .. TW_ROWS ; [-1] and [TW_ROWS] stay zero
  float* Bs = smem + (TW_ROWS + 2) * BB_AP;               // [9][5][128] the first layer's lookup table
  float* psum = Bs + 9 * 5 * BB_C;                        // [4][TW_ROWS]
  float* rstat = psum + 8 * TW_ROWS;                      // [TW_ROWS]
  int* toks = reinterpret_cast<int*>(rstat + TW_ROWS);    // [TW_ROWS]
  int* rpos = toks + TW_ROWS;                             // [TW_ROWS] position of a tile row inside its sequence
  int* sdil = rpos + TW_ROWS;                             // [BB_MAXL + 1]
  int* sched = sdil + BB_MAXL + 1;                        // [(nl + 1) * 36]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w & 3, rh = w >> 2;
  const int j = lane & 15, g = lane >> 4;
  const int col0 = 32 * cg + j;                           // this lane's columns: col0 and col0 + 16
  const int L = a.L;
  const int nvalid = a.count ? __builtin_amdgcn_readfirstlane(*a.count) : a.n;
  int seq0 = blockIdx.x, spt = 1;                         // this tile's first sequence and how many it takes
  if (!SPT1) {
    if ((int)blockIdx.x >= nvalid) return;                // a tile takes at least one sequence: no plan needed to know this one is empty
    SvddTilePlan pl = a.plan;
    if (a.auto_spt) pl = svdd_plan_tiles(nvalid, L, a.ncu, 9);
    svdd_plan_tile(pl, (int)blockIdx.x, seq0, spt);
    seq0 = __builtin_amdgcn_readfirstlane(seq0);
    spt = __builtin_amdgcn_readfirstlane(spt);
  }
  const int tile_rows = spt * L;
  if (seq0 >= nvalid) return;
  // Short tiles (round 6): a tile of nt < 13 row tiles leaves some of a wave's 7 / 6 row-tile slots dead — and a dead slot still cost
  // its fragment request, liveness test and wait in every (chunk, tap) entry, and its rows their LayerNorm passes (a 4-row-tile
  // workgroup: 10 dead slots per entry, 0.73 pipe occupancy in the loop, 14 % of its time in LayerNorm code for rows that do not exist;
  // profiles/r06_bb_small_tile_phases.txt). nro = the row-tile slots this wave really owns; the entry bodies come in three lengths
  // (7 / 4 / 2 slots) and the LayerNorm passes skip the slots beyond nro. Same operations on every live row: same bits.
  const int nt = SPT1 ? TW_RT : (tile_rows + 15) >> 4;
  const int nro = SPT1 ? 7 : __builtin_amdgcn_readfirstlane((nt - rh + 1) >> 1);
  const int body = SPT1 ? 7 : (nt <= 4 ? 2 : nt <= 8 ? 4 : 7);
#define BB_OWN(R) (SPT1 || (R) < nro)
  const int nl = a.nl;
  const int it_end = (nl + 1) * 36;
  constexpr int NR = 7;                                   // owned row tiles rh + 2 r (r = 6 only for rh = 0)

  // Several sequences per tile (!SPT1, L <= 104; round 5): the `spt` sequences are INTERLEAVED position-major — tile row
  // e = position * spt + q holds position e / spt of sequence seq0 + q. A dilated tap then is a row offset of spt * (t - 4) * dil that
  // leaves the tile exactly when the position leaves the sequence: the tile behaves like ONE sequence of spt * L rows with every
  // dilation multiplied by spt — clamped addressing and tile-granular liveness as for L = 200, instead of a position test per
  // fragment row and row tiles kept live by a few rows of each stacked sequence (the stacked layout issued 425 tile-taps per five
  // layers for 372 useful ones). Same products in the same order per output element: same bits.
  const int il = SPT1 ? 1 : spt;
  for (int e = tid; e < TW_ROWS; e += 512) {
    int tk = -1;
    const int q = SPT1 ? 0 : e % il, pos = SPT1 ? e : e / il;
    if (e < tile_rows && seq0 + q < nvalid) {
      const int64_t sq = a.row_idx ? (int64_t)a.row_idx[seq0 + q] : (int64_t)(seq0 + q);
      tk = a.x[sq * L + pos];
    }
    toks[e] = tk;
    rpos[e] = e < tile_rows ? pos : -(1 << 20);
  }
  for (int e = tid; e < BB_AP; e += 512) { smem[e] = 0.0f; img[TW_ROWS * BB_AP + e] = 0.0f; }
  if (!SPT1)                                              // rows of the slots nobody owns: zero once, never written again (taps read them as padding)
    for (int e = 16 * nt * BB_AP + tid; e < TW_ROWS * BB_AP; e += 512) img[e] = 0.0f;
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < BB_MAXL; ++i) sdil[i] = a.dil[i];
    sdil[BB_MAXL] = 1;
  }
  for (int e = tid; e < 9 * 5 * BB_C; e += 512) Bs[e] = a.table0[e];
  __syncthreads();
  // Schedule: index k = (layer*4 + chunk)*9 + tap. sched[k] = 0 for a tap that only sees zero padding, else
  //   bits 0-12: live row tiles ; 13-14 chunk ; 15-18 tap ; 19-28 index of the next live k.
  for (int k = tid; k < it_end; k += 512) {
    auto entry = [&](int kk) {
      const int layer = kk / 36, t = kk % 9;
      if (layer >= nl) return t == 4 ? 0x1fff : 0;
      const int d = (t - 4) * sdil[layer] * il;           // in tile rows (interleaved sequences: il rows per position)
      const int lo = d < 0 ? -d : 0, hi = d > 0 ? tile_rows - d : tile_rows;
      if (lo >= hi) return 0;
      int m = 0;
      for (int r = 0; r < TW_RT; ++r) if (lo < 16 * r + 16 && hi > 16 * r) m |= 1 << r;
      return m;
    };
    const int m = entry(k);
    int nx = k + 1;
    while (nx < it_end && entry(nx) == 0) ++nx;
    sched[k] = m ? (m | ((k % 36) / 9) << 13 | (k % 9) << 15 | nx << 19) : 0;
  }

  // ---- first layer: f[row][col] = relu(b + sum_t table[t][tok[row + t - 4]][col])   (dnaconv.py:177,184)
  f32x4 f[NR][2], acc[NR][2];
  unsigned long long relu_bits = 0ull;                    // SAVE: this lane's ReLU decisions of the current layer
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = col0 + 16 * ct;
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
            const int tk = (p >= 0 && p < L) ? toks[row + (t - 4) * il] : -1;
            if (tk >= 0) v += Bs[(t * 5 + tk) * BB_C + col];
          }
        }
        f[r][ct][e] = row < tile_rows ? fmaxf(v, 0.0f) : 0.0f;
        if (SAVE && row < tile_rows && v > 0.0f) relu_bits |= 1ull << ((r * 2 + ct) * 4 + e);
      }
  }
  if (SAVE) sv.mask[(size_t)blockIdx.x * (nl + 2) * 512 + tid] = relu_bits;
  __syncthreads();                                        // sched is visible

  // A operand addressing: this lane feeds row 16 (rh + 2 r) + j of its tiles, channels 32 c + 8 g .. + 8
  typedef __attribute__((address_space(3))) f32x4 LdsF4;
  const int img_lds = (int)(unsigned)(size_t)(const __attribute__((address_space(3))) float*)img;              // LDS byte address of img
  const int arow0 = img_lds + ((16 * rh + j) * BB_AP + 8 * g) * 4;  // LDS byte address of (row 16 rh + j, col 8 g) of img
  const int a_lo = arow0 - (16 * rh + j + 1) * BB_AP * 4; // row -1
  const int a_hi = arow0 + (TW_ROWS - 16 * rh - j) * BB_AP * 4;     // row TW_ROWS

  // Weight stream: each lane reads W[col0][8 g ..] and W[col0 + 16][8 g ..] of the (layer, chunk, tap) tile straight
  // from L2 into the B-operand registers, one tile ahead (no LDS staging, no barrier inside a layer).
  const float* wsrc = a.tiles + col0 * CH + 8 * g;
  auto tile_of = [&](int k) { return k < nl * 36 ? k : nl * 36 + (k - nl * 36) / 9; };
  int it = 0;
  while (it < it_end && sched[it] == 0) ++it;
  it = __builtin_amdgcn_readfirstlane(it);
  int en = __builtin_amdgcn_readfirstlane(sched[it]);
  // two weight-fragment sets that trade roles from entry to entry (round 4d: with ONE set refilled in place the current tile
  // had to be copied out first, 8 v_mov_b64 at each end of an entry, and VALU work does not run under fp32 MFMAs on this SIMD)
  float4 bA[4], bB[4];
  {
    const float* src = wsrc + (size_t)tile_of(it) * BB_C * CH;
    bA[0] = *reinterpret_cast<const float4*>(src);
    bA[1] = *reinterpret_cast<const float4*>(src + 4);
    bA[2] = *reinterpret_cast<const float4*>(src + 16 * CH);
    bA[3] = *reinterpret_cast<const float4*>(src + 16 * CH + 4);
    bB[0] = bA[0]; bB[1] = bA[1]; bB[2] = bA[2]; bB[3] = bA[3];
  }

  for (int layer = 0; layer <= nl; ++layer) {             // layer == nl: the first 1x1 conv of final_conv
    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;
    if (layer < nl) {
      const float tb0 = vl[BB_C + col0], tb1 = vl[BB_C + col0 + 16];
      // pass 1: row means (this lane's two values of a row first, then the 16 lanes of the DPP row)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (!BB_OWN(r)) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          const float sm = group16_sum((f[r][0][e] + tb0) + (f[r][1][e] + tb1));
          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sm;
        }
      }
      __syncthreads();
      if (tid < TW_ROWS)
        rstat[tid] = ((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) * (1.0f / BB_C);
      __syncthreads();
      // pass 2: centred second moment
      // the four rows of a tile that a lane owns are adjacent: one 16-byte read per tile, issued one tile ahead (a read
      // placed next to its use would pay a full LDS round trip per row: the stores in between pin it in place)
      float4 st4 = *reinterpret_cast<const float4*>(rstat + min(16 * rh + 4 * g, TW_ROWS - 4));
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float4 cur4 = st4;
        if (r + 1 < NR) st4 = *reinterpret_cast<const float4*>(rstat + min(16 * (rh + 2 * r + 2) + 4 * g, TW_ROWS - 4));
        const float mean4[4] = {cur4.x, cur4.y, cur4.z, cur4.w};
        if (!BB_OWN(r)) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          const float mean = mean4[e];
          const float d0 = f[r][0][e] + tb0 - mean, d1 = f[r][1][e] + tb1 - mean;
          acc[r][0][e] = d0; acc[r][1][e] = d1;           // keep the centred values
          const float sq = group16_sum(d0 * d0 + d1 * d1);
          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sq;
        }
      }
      __syncthreads();
      if (tid < TW_ROWS)
        rstat[tid] = rsqrtf(((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) *
                            (1.0f / BB_C) + 1e-5f);
      __syncthreads();
      const float gm0 = vl[2 * BB_C + col0], gm1 = vl[2 * BB_C + col0 + 16];
      const float bt0 = vl[3 * BB_C + col0], bt1 = vl[3 * BB_C + col0 + 16];
      st4 = *reinterpret_cast<const float4*>(rstat + min(16 * rh + 4 * g, TW_ROWS - 4));
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float4 cur4 = st4;
        if (r + 1 < NR) st4 = *reinterpret_cast<const float4*>(rstat + min(16 * (rh + 2 * r + 2) + 4 * g, TW_ROWS - 4));
        const float rs4[4] = {cur4.x, cur4.y, cur4.z, cur4.w};
        if (!BB_OWN(r)) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) {
            const float rs = rs4[e];
            img[row * BB_AP + col0] = row < tile_rows ? acc[r][0][e] * rs * gm0 + bt0 : 0.0f;
            img[row * BB_AP + col0 + 16] = row < tile_rows ? acc[r][1][e] * rs * gm1 + bt1 : 0.0f;
            if (SAVE) {
              float* xh = sv.xhat + (((size_t)blockIdx.x * nl + layer) * 56 + (r * 2) * 4 + e) * 512 + tid;
              xh[0] = acc[r][0][e] * rs;
              xh[4 * 512] = acc[r][1][e] * rs;
            }
          }
        }
      }
      if (SAVE && tid < TW_ROWS) sv.rstd[((size_t)blockIdx.x * nl + layer) * TW_ROWS + tid] = rstat[tid];
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (!BB_OWN(r)) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) { img[row * BB_AP + col0] = f[r][0][e]; img[row * BB_AP + col0 + 16] = f[r][1][e]; }
        }
      }
    }
    // ---- implicit GEMM over (chunk, live tap)
    const float bl0 = vl[col0], bl1 = vl[col0 + 16];
#pragma unroll
    for (int r = 0; r < NR; ++r) { acc[r][0] = f32x4{bl0, bl0, bl0, bl0}; acc[r][1] = f32x4{bl1, bl1, bl1, bl1}; }
    const int dil = __builtin_amdgcn_readfirstlane(sdil[layer < nl ? layer : BB_MAXL]) * il;   // in tile rows
    const int layer_end = (layer + 1) * 36;
    __syncthreads();                                      // the image is complete
    // A fragments of row tiles 0 and 1 of an entry are requested during the LAST MFMA groups of the entry before it (first
    // entry of a layer: right here), so that the matrix pipe does not run dry at every (chunk, tap) boundary while both waves
    // of a SIMD wait for their first LDS reads (round 3; same bits: only the loads move). A wave with 7 row tiles ends an
    // entry on the fragment set it began with, so its two sets trade roles from entry to entry (two copies of the body).
    // (VALU work does not run under fp32 MFMAs on this SIMD: the clamp is ONE v_med3_i32, and the fragment is read through an
    //  LDS address kept as an integer that already contains the image's base — as "imgb + o_" every request carried a
    //  v_add_u32 of the dynamic-LDS symbol, which is 0)
#define B2_ALOAD(R, V, DELTA, COFF, DBYTES)                                                                  \
      { int o_;                                                                                              \
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(o_) : "v"(arow0 + (DBYTES) + (R) * (32 * BB_AP * 4)), "v"(a_lo + (COFF)), "v"(a_hi + (COFF))); \
        const LdsF4* ap_ = reinterpret_cast<const LdsF4*>(o_);                                               \
        const f32x4 t0_ = ap_[0], t1_ = ap_[1];                                                              \
        V[0] = make_float4(t0_[0], t0_[1], t0_[2], t0_[3]); V[1] = make_float4(t1_[0], t1_[1], t1_[2], t1_[3]); }
#define B2_WAIT(NOUT) __builtin_amdgcn_s_waitcnt(0xC07F | ((NOUT) << 8));
#define B2_MM(R, U, NOUT)                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      B2_WAIT(NOUT)                                                                                          \
      if (live & (1 << (2 * (R)))) {                                                                         \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].x, bf0[4 * q], acc[R][0], 0, 0, 0);          \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].x, bf1[4 * q], acc[R][1], 0, 0, 0);          \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].y, bf0[4 * q + 1], acc[R][0], 0, 0, 0);      \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].y, bf1[4 * q + 1], acc[R][1], 0, 0, 0);      \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].z, bf0[4 * q + 2], acc[R][0], 0, 0, 0);      \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].z, bf1[4 * q + 2], acc[R][1], 0, 0, 0);      \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].w, bf0[4 * q + 3], acc[R][0], 0, 0, 0);      \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].w, bf1[4 * q + 3], acc[R][1], 0, 0, 0);      \
        }                                                                                                    \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);
    // entry parameters from a schedule word
#define B2_PARAMS(EN, DELTA, COFF, DBYTES)                                                                   \
      const int DELTA = ((((EN) >> 15) & 15) - 4) * dil;                                                     \
      const int COFF = (((EN) >> 13) & 3) * (CH * 4);                                                        \
      const int DBYTES = DELTA * (BB_AP * 4) + COFF;
    // one (chunk, tap) entry; UA holds (or is receiving) row tile 0, UB row tile 1
#define B2_ENTRY(UA, UB, BC, BN)                                                                             \
    { const int nxt = en >> 19;                                                                              \
      const int en_next_v = sched[nxt < it_end ? nxt : it];                                                  \
      const float bf0[8] = {BC[0].x, BC[0].y, BC[0].z, BC[0].w, BC[1].x, BC[1].y, BC[1].z, BC[1].w};         \
      const float bf1[8] = {BC[2].x, BC[2].y, BC[2].z, BC[2].w, BC[3].x, BC[3].y, BC[3].z, BC[3].w};         \
      { /* the next entry's tile into the idle set; after the last entry: this entry's again (never used) — unconditional, \
           so that the compiler's vmcnt model has one path */                                                \
        const float* src = wsrc + (size_t)tile_of(nxt < it_end ? nxt : it) * BB_C * CH;                      \
        BN[0] = *reinterpret_cast<const float4*>(src);                                                       \
        BN[1] = *reinterpret_cast<const float4*>(src + 4);                                                   \
        BN[2] = *reinterpret_cast<const float4*>(src + 16 * CH);                                             \
        BN[3] = *reinterpret_cast<const float4*>(src + 16 * CH + 4);                                         \
      }                                                                                                      \
      B2_PARAMS(en, delta, coff, dbytes)                                                                     \
      const int live = en >> rh;                          /* bit 2 r = owned tile r */                       \
      /* (after a layer's last entry the "next entry" loads below read the dying image with the wrong dilation: clamped \
         addresses, values never used — the layer prologue reloads both sets; cheaper than a second copy of the MFMA groups) */ \
      B2_MM(0, UA, 3)                                     /* in flight: tile 1 (2 reads) + the schedule word */ \
      B2_ALOAD(2, UA, delta, coff, dbytes)                                                                   \
      B2_MM(1, UB, 2)                                                                                        \
      const int en2 = __builtin_amdgcn_readfirstlane(en_next_v);                                             \
      B2_PARAMS(en2, delta2, coff2, dbytes2)                                                                 \
      B2_ALOAD(3, UB, delta, coff, dbytes)                                                                   \
      B2_MM(2, UA, 2)                                                                                        \
      B2_ALOAD(4, UA, delta, coff, dbytes)                                                                   \
      B2_MM(3, UB, 2)                                                                                        \
      B2_ALOAD(5, UB, delta, coff, dbytes)                                                                   \
      B2_MM(4, UA, 2)                                                                                        \
      if (rh == 0) {                                                                                         \
        B2_ALOAD(6, UA, delta, coff, dbytes)                                                                 \
        B2_MM(5, UB, 2)                                                                                      \
        B2_ALOAD(0, UB, delta2, coff2, dbytes2)           /* next entry: tile 0 in UB, tile 1 in UA */       \
        B2_MM(6, UA, 2)                                                                                      \
        B2_ALOAD(1, UA, delta2, coff2, dbytes2)                                                              \
      } else {                                                                                               \
        B2_ALOAD(0, UA, delta2, coff2, dbytes2)                                                              \
        B2_MM(5, UB, 2)                                                                                      \
        B2_ALOAD(1, UB, delta2, coff2, dbytes2)                                                              \
      }                                                                                                      \
      it = nxt;                                                                                              \
      en = en2; }
    // the same entry for a wave with at most FOUR / TWO owned row tiles (tiles of <= 8 / <= 4 row tiles): both waves of a SIMD take the
    // same even number of slots, so the fragment sets keep their roles from entry to entry
#define B2_ENTRY_HEAD(BC, BN)                                                                                \
      const int nxt = en >> 19;                                                                              \
      const int en_next_v = sched[nxt < it_end ? nxt : it];                                                  \
      const float bf0[8] = {BC[0].x, BC[0].y, BC[0].z, BC[0].w, BC[1].x, BC[1].y, BC[1].z, BC[1].w};         \
      const float bf1[8] = {BC[2].x, BC[2].y, BC[2].z, BC[2].w, BC[3].x, BC[3].y, BC[3].z, BC[3].w};         \
      { const float* src = wsrc + (size_t)tile_of(nxt < it_end ? nxt : it) * BB_C * CH;                      \
        BN[0] = *reinterpret_cast<const float4*>(src);                                                       \
        BN[1] = *reinterpret_cast<const float4*>(src + 4);                                                   \
        BN[2] = *reinterpret_cast<const float4*>(src + 16 * CH);                                             \
        BN[3] = *reinterpret_cast<const float4*>(src + 16 * CH + 4);                                         \
      }                                                                                                      \
      B2_PARAMS(en, delta, coff, dbytes)                                                                     \
      const int live = en >> rh;
#define B2_ENTRY4(UA, UB, BC, BN)                                                                            \
    { B2_ENTRY_HEAD(BC, BN)                                                                                  \
      B2_MM(0, UA, 3)                                                                                        \
      B2_ALOAD(2, UA, delta, coff, dbytes)                                                                   \
      B2_MM(1, UB, 2)                                                                                        \
      const int en2 = __builtin_amdgcn_readfirstlane(en_next_v);                                             \
      B2_PARAMS(en2, delta2, coff2, dbytes2)                                                                 \
      B2_ALOAD(3, UB, delta, coff, dbytes)                                                                   \
      B2_MM(2, UA, 2)                                                                                        \
      B2_ALOAD(0, UA, delta2, coff2, dbytes2)                                                                \
      B2_MM(3, UB, 2)                                                                                        \
      B2_ALOAD(1, UB, delta2, coff2, dbytes2)                                                                \
      it = nxt;                                                                                              \
      en = en2; }
#define B2_ENTRY2(UA, UB, BC, BN)                                                                            \
    { B2_ENTRY_HEAD(BC, BN)                                                                                  \
      B2_MM(0, UA, 3)                                                                                        \
      const int en2 = __builtin_amdgcn_readfirstlane(en_next_v);                                             \
      B2_PARAMS(en2, delta2, coff2, dbytes2)                                                                 \
      B2_ALOAD(0, UA, delta2, coff2, dbytes2)                                                                \
      B2_MM(1, UB, 2)                                                                                        \
      B2_ALOAD(1, UB, delta2, coff2, dbytes2)                                                                \
      it = nxt;                                                                                              \
      en = en2; }
    float4 ua[2], ub[2];
    if (it < layer_end) {
      B2_PARAMS(en, delta0, coff0, dbytes0)
      B2_ALOAD(0, ua, delta0, coff0, dbytes0)
      B2_ALOAD(1, ub, delta0, coff0, dbytes0)
    }
    // entries come in fours (the chunks of a tap), so a layer always ends on the second body and the sets keep their roles
    // from layer to layer; the odd exit is there for the bookkeeping only
    bool odd = false;
    if (!SPT1 && body == 4) {
      while (it < layer_end) {
        B2_ENTRY4(ua, ub, bA, bB)
        if (it >= layer_end) { odd = true; break; }
        B2_ENTRY4(ua, ub, bB, bA)
      }
    } else if (!SPT1 && body == 2) {
      while (it < layer_end) {
        B2_ENTRY2(ua, ub, bA, bB)
        if (it >= layer_end) { odd = true; break; }
        B2_ENTRY2(ua, ub, bB, bA)
      }
    } else if (rh == 0) {
      while (it < layer_end) {
        B2_ENTRY(ua, ub, bA, bB)
        if (it >= layer_end) { odd = true; break; }
        B2_ENTRY(ub, ua, bB, bA)
      }
    } else {
      while (it < layer_end) {
        B2_ENTRY(ua, ub, bA, bB)
        if (it >= layer_end) { odd = true; break; }
        B2_ENTRY(ua, ub, bB, bA)
      }
    }
    if (odd) { bA[0] = bB[0]; bA[1] = bB[1]; bA[2] = bB[2]; bA[3] = bB[3]; }
#undef B2_ENTRY
#undef B2_ENTRY4
#undef B2_ENTRY2
#undef B2_ENTRY_HEAD
#undef B2_PARAMS
#undef B2_MM
#undef B2_WAIT
#undef B2_ALOAD
    __syncthreads();                                      // every wave is done reading the image
    if (SAVE) {
      relu_bits = 0ull;
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (acc[r][ct][e] > 0.0f) relu_bits |= 1ull << ((r * 2 + ct) * 4 + e);
      sv.mask[((size_t)blockIdx.x * (nl + 2) + 1 + layer) * 512 + tid] = relu_bits;
    }
    if (layer < nl) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (!BB_OWN(r)) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e) f[r][ct][e] = fmaxf(acc[r][ct][e], 0.0f) + f[r][ct][e];   // relu(conv + b) + f
      }
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (!BB_OWN(r)) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) {
            img[row * BB_AP + col0] = fmaxf(acc[r][0][e], 0.0f);                                   // relu(W1 f + b1)
            img[row * BB_AP + col0 + 16] = fmaxf(acc[r][1][e], 0.0f);
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- last 1x1 conv 128 -> 5: one (row, class) dot product per thread iteration
  for (int e = tid; e < tile_rows * 5; e += 512) {
    const int row = e / 5, v = e - 5 * row;
    const int q = SPT1 ? 0 : row % il, pos = SPT1 ? row : row / il;   // tile row -> (sequence seq0 + q, position)
    if (seq0 + q >= nvalid) continue;
    const float* hr = img + row * BB_AP;
    const float* wv = a.w2 + v * BB_C;
    float sm = a.w2[5 * BB_C + v];
#pragma unroll 8
    for (int k = 0; k < BB_C; ++k) sm += hr[k] * wv[k];
    const int64_t sq = (a.row_idx && a.out_scatter) ? (int64_t)a.row_idx[seq0 + q] : (int64_t)(seq0 + q);
    a.out[(sq * L + pos) * 5 + v] = sm;
  }
}
#undef BB_OWN

// ------------------------------------------- gradient of the backbone with respect to its one-hot input, ONE launch (round 5) ----
// The gradient-guidance baseline (DPS, reference diffusion_gosai.py:1321-1330 through models/dnaconv.py:212-247) needs
// d loss / d onehot(x_t) through the whole backbone; the weights are frozen. backbone_kernel<true, true> is the forward (the
// inference kernel's bits) that also leaves, per layer, x-hat, 1 / sigma and the ReLU decisions in its lane-private layout
// (BackboneSave); this kernel walks the layers in reverse with the SAME work split — wave (cg, rh) owns columns 32 cg .. + 32 of
// the row tiles rh + 2 r, the residual GRADIENT stream G lives in the registers the forward kept f in — so that every saved value
// comes back to the lane that wrote it in one coalesced load:
//     G1 = (dlogits W2) * relu'(final 1x1)                                    VALU, 5 MACs per element
//     G  = G1 W1                                                              1x1 conv with W1^T: the MFMA loop, taps = {4}
//     for i = nl - 1 .. 0:   dhn = conv9_dil_i^T (G * relu'_i)                the SAME implicit-GEMM loop on W_i with flipped taps and
//                                                                             swapped channel axes (packed by the host)
//                            G  += LayerNorm'(dhn ; xhat_i, rstd_i, gamma_i)   rstd (t - mean(t) - xhat mean(t xhat)), t = gamma dhn
//     dx[p][c] = sum_t sum_co W_first[co][c][t] (G * relu'_first)[p - (t - 4)][co]
// Tiles arrive in PROCESSING order: [4][128][32] of W1^T, then layer nl - 1, nl - 2, ... 0 as [4 chunks][9 taps][128][32].
struct BackboneGradArgs {
  const float* dlogits;    // [n][L][5]
  const float* tiles;      // see above
  const float* gamma;      // [nl][128], layer order
  const float* w2;         // [5][128]
  const float* table0;     // [9][5][128] = W_first[co][c][t] -> [t][c][co]
  float* dx;               // [n][L][5]
  int n, L, nl;
  int dil[BB_MAXL];        // layer order
};

__global__ __launch_bounds__(512, 2) void backbone_grad_kernel(BackboneGradArgs a, BackboneSave sv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img = smem + BB_AP;                              // rows -1 .. TW_ROWS ; [-1] and [TW_ROWS] stay zero
  float* Bs = smem + (TW_ROWS + 2) * BB_AP;               // [9][5][128]
  float* psum = Bs + 9 * 5 * BB_C;                        // [8][TW_ROWS]: partial sums of t (0..3) and of t xhat (4..7); first: dlogits
  float* rstat = psum + 8 * TW_ROWS;                      // [2][TW_ROWS]
  int* sdil = reinterpret_cast<int*>(rstat + 2 * TW_ROWS);  // [BB_MAXL + 1] dilation of processing step s (step 0 = the 1x1)
  int* sched = sdil + BB_MAXL + 1;                        // [(nl + 1) * 36]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w & 3, rh = w >> 2;
  const int j = lane & 15, g = lane >> 4;
  const int col0 = 32 * cg + j;
  const int L = a.L, nl = a.nl;
  const int tile_rows = L;
  const int64_t row0 = (int64_t)blockIdx.x * L;
  const int it_end = (nl + 1) * 36;
  constexpr int NR = 7;

  for (int e = tid; e < BB_AP; e += 512) { smem[e] = 0.0f; img[TW_ROWS * BB_AP + e] = 0.0f; }
  if (tid <= nl) sdil[tid] = tid == 0 ? 1 : a.dil[nl - tid];
  for (int e = tid; e < 9 * 5 * BB_C; e += 512) Bs[e] = a.table0[e];
  for (int e = tid; e < TW_ROWS * 5; e += 512) psum[e] = e < L * 5 ? a.dlogits[row0 * 5 + e] : 0.0f;
  __syncthreads();
  // Schedule: index k = (step*4 + chunk)*9 + tap ; the same word format as backbone_kernel's
  for (int k = tid; k < it_end; k += 512) {
    auto entry = [&](int kk) {
      const int step = kk / 36, t = kk % 9;
      if (step == 0) return t == 4 ? 0x1fff : 0;
      const int d = (t - 4) * sdil[step];
      const int lo = d < 0 ? -d : 0, hi = d > 0 ? L - d : L;
      if (lo >= hi) return 0;
      int m = 0;
      for (int r = 0; r < TW_RT; ++r) if (lo < 16 * r + 16 && hi > 16 * r) m |= 1 << r;
      return m;
    };
    const int m = entry(k);
    int nx = k + 1;
    while (nx < it_end && entry(nx) == 0) ++nx;
    sched[k] = m ? (m | ((k % 36) / 9) << 13 | (k % 9) << 15 | nx << 19) : 0;
  }

  // ---- G1 = (dlogits W2) * relu'(final 1x1) -> the image
  f32x4 G[NR][2], acc[NR][2];
  {
    const unsigned long long mk = sv.mask[((size_t)blockIdx.x * (nl + 2) + nl + 1) * 512 + tid];
    float w2a[5], w2b[5];
#pragma unroll
    for (int v = 0; v < 5; ++v) { w2a[v] = a.w2[v * BB_C + col0]; w2b[v] = a.w2[v * BB_C + col0 + 16]; }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * (rh + 2 * r) + 4 * g + e;
        if (row < TW_ROWS) {
          float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
          for (int v = 0; v < 5; ++v) { const float dl = psum[row * 5 + v]; s0 += dl * w2a[v]; s1 += dl * w2b[v]; }
          img[row * BB_AP + col0] = ((mk >> ((r * 2) * 4 + e)) & 1ull) && row < tile_rows ? s0 : 0.0f;
          img[row * BB_AP + col0 + 16] = ((mk >> ((r * 2 + 1) * 4 + e)) & 1ull) && row < tile_rows ? s1 : 0.0f;
        }
      }
  }
  __syncthreads();                                        // sched and the image are visible

  typedef __attribute__((address_space(3))) f32x4 LdsF4;
  const int img_lds = (int)(unsigned)(size_t)(const __attribute__((address_space(3))) float*)img;
  const int arow0 = img_lds + ((16 * rh + j) * BB_AP + 8 * g) * 4;
  const int a_lo = arow0 - (16 * rh + j + 1) * BB_AP * 4;
  const int a_hi = arow0 + (TW_ROWS - 16 * rh - j) * BB_AP * 4;

  const float* wsrc = a.tiles + col0 * CH + 8 * g;
  auto tile_of = [&](int k) { return k < 36 ? k / 9 : 4 + (k - 36); };
  int it = 0;
  while (it < it_end && sched[it] == 0) ++it;
  it = __builtin_amdgcn_readfirstlane(it);
  int en = __builtin_amdgcn_readfirstlane(sched[it]);
  float4 bA[4], bB[4];
  {
    const float* src = wsrc + (size_t)tile_of(it) * BB_C * CH;
    bA[0] = *reinterpret_cast<const float4*>(src);
    bA[1] = *reinterpret_cast<const float4*>(src + 4);
    bA[2] = *reinterpret_cast<const float4*>(src + 16 * CH);
    bA[3] = *reinterpret_cast<const float4*>(src + 16 * CH + 4);
    bB[0] = bA[0]; bB[1] = bA[1]; bB[2] = bA[2]; bB[3] = bA[3];
  }

  for (int step = 0; step <= nl; ++step) {
    const int layer = nl - step;                          // step >= 1: the conv layer whose transpose this step applies
    if (step > 0) {
      // the image of this step: G * relu'(conv_layer), rows beyond the sequence zero
      const unsigned long long mk = sv.mask[((size_t)blockIdx.x * (nl + 2) + 1 + layer) * 512 + tid];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          if (row < TW_ROWS) {
            img[row * BB_AP + col0] = ((mk >> ((r * 2) * 4 + e)) & 1ull) && row < tile_rows ? G[r][0][e] : 0.0f;
            img[row * BB_AP + col0 + 16] = ((mk >> ((r * 2 + 1) * 4 + e)) & 1ull) && row < tile_rows ? G[r][1][e] : 0.0f;
          }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) { acc[r][0] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; acc[r][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    const int dil = __builtin_amdgcn_readfirstlane(sdil[step]);
    const int layer_end = (step + 1) * 36;
    __syncthreads();                                      // the image is complete
#define B2_ALOAD(R, V, DELTA, COFF, DBYTES)                                                                  \
      { int o_;                                                                                              \
        asm("v_med3_i32 %0, %1, %2, %3" : "=v"(o_) : "v"(arow0 + (DBYTES) + (R) * (32 * BB_AP * 4)), "v"(a_lo + (COFF)), "v"(a_hi + (COFF))); \
        const LdsF4* ap_ = reinterpret_cast<const LdsF4*>(o_);                                               \
        const f32x4 t0_ = ap_[0], t1_ = ap_[1];                                                              \
        V[0] = make_float4(t0_[0], t0_[1], t0_[2], t0_[3]); V[1] = make_float4(t1_[0], t1_[1], t1_[2], t1_[3]); }
#define B2_WAIT(NOUT) __builtin_amdgcn_s_waitcnt(0xC07F | ((NOUT) << 8));
#define B2_MM(R, U, NOUT)                                                                                    \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      B2_WAIT(NOUT)                                                                                          \
      if (live & (1 << (2 * (R)))) {                                                                         \
        _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                      \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].x, bf0[4 * q], acc[R][0], 0, 0, 0);          \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].x, bf1[4 * q], acc[R][1], 0, 0, 0);          \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].y, bf0[4 * q + 1], acc[R][0], 0, 0, 0);      \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].y, bf1[4 * q + 1], acc[R][1], 0, 0, 0);      \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].z, bf0[4 * q + 2], acc[R][0], 0, 0, 0);      \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].z, bf1[4 * q + 2], acc[R][1], 0, 0, 0);      \
          acc[R][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].w, bf0[4 * q + 3], acc[R][0], 0, 0, 0);      \
          acc[R][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[q].w, bf1[4 * q + 3], acc[R][1], 0, 0, 0);      \
        }                                                                                                    \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);
#define B2_PARAMS(EN, DELTA, COFF, DBYTES)                                                                   \
      const int DELTA = ((((EN) >> 15) & 15) - 4) * dil;                                                     \
      const int COFF = (((EN) >> 13) & 3) * (CH * 4);                                                        \
      const int DBYTES = DELTA * (BB_AP * 4) + COFF;
#define B2_ENTRY(UA, UB, BC, BN)                                                                             \
    { const int nxt = en >> 19;                                                                              \
      const int en_next_v = sched[nxt < it_end ? nxt : it];                                                  \
      const float bf0[8] = {BC[0].x, BC[0].y, BC[0].z, BC[0].w, BC[1].x, BC[1].y, BC[1].z, BC[1].w};         \
      const float bf1[8] = {BC[2].x, BC[2].y, BC[2].z, BC[2].w, BC[3].x, BC[3].y, BC[3].z, BC[3].w};         \
      { const float* src = wsrc + (size_t)tile_of(nxt < it_end ? nxt : it) * BB_C * CH;                      \
        BN[0] = *reinterpret_cast<const float4*>(src);                                                       \
        BN[1] = *reinterpret_cast<const float4*>(src + 4);                                                   \
        BN[2] = *reinterpret_cast<const float4*>(src + 16 * CH);                                             \
        BN[3] = *reinterpret_cast<const float4*>(src + 16 * CH + 4);                                         \
      }                                                                                                      \
      B2_PARAMS(en, delta, coff, dbytes)                                                                     \
      const int live = en >> rh;                                                                             \
      B2_MM(0, UA, 3)                                                                                        \
      B2_ALOAD(2, UA, delta, coff, dbytes)                                                                   \
      B2_MM(1, UB, 2)                                                                                        \
      const int en2 = __builtin_amdgcn_readfirstlane(en_next_v);                                             \
      B2_PARAMS(en2, delta2, coff2, dbytes2)                                                                 \
      B2_ALOAD(3, UB, delta, coff, dbytes)                                                                   \
      B2_MM(2, UA, 2)                                                                                        \
      B2_ALOAD(4, UA, delta, coff, dbytes)                                                                   \
      B2_MM(3, UB, 2)                                                                                        \
      B2_ALOAD(5, UB, delta, coff, dbytes)                                                                   \
      B2_MM(4, UA, 2)                                                                                        \
      if (rh == 0) {                                                                                         \
        B2_ALOAD(6, UA, delta, coff, dbytes)                                                                 \
        B2_MM(5, UB, 2)                                                                                      \
        B2_ALOAD(0, UB, delta2, coff2, dbytes2)                                                              \
        B2_MM(6, UA, 2)                                                                                      \
        B2_ALOAD(1, UA, delta2, coff2, dbytes2)                                                              \
      } else {                                                                                               \
        B2_ALOAD(0, UA, delta2, coff2, dbytes2)                                                              \
        B2_MM(5, UB, 2)                                                                                      \
        B2_ALOAD(1, UB, delta2, coff2, dbytes2)                                                              \
      }                                                                                                      \
      it = nxt;                                                                                              \
      en = en2; }
    float4 ua[2], ub[2];
    if (it < layer_end) {
      B2_PARAMS(en, delta0, coff0, dbytes0)
      B2_ALOAD(0, ua, delta0, coff0, dbytes0)
      B2_ALOAD(1, ub, delta0, coff0, dbytes0)
    }
    bool odd = false;
    if (rh == 0) {
      while (it < layer_end) {
        B2_ENTRY(ua, ub, bA, bB)
        if (it >= layer_end) { odd = true; break; }
        B2_ENTRY(ub, ua, bB, bA)
      }
    } else {
      while (it < layer_end) {
        B2_ENTRY(ua, ub, bA, bB)
        if (it >= layer_end) { odd = true; break; }
        B2_ENTRY(ua, ub, bB, bA)
      }
    }
    if (odd) { bA[0] = bB[0]; bA[1] = bB[1]; bA[2] = bB[2]; bA[3] = bB[3]; }
#undef B2_ENTRY
#undef B2_PARAMS
#undef B2_MM
#undef B2_WAIT
#undef B2_ALOAD
    __syncthreads();                                      // every wave is done reading the image
    if (step == 0) {
#pragma unroll
      for (int r = 0; r < NR; ++r) { G[r][0] = acc[r][0]; G[r][1] = acc[r][1]; }
    } else {
      // LayerNorm backward: t = gamma dhn ; dh = rstd (t - mean(t) - xhat mean(t xhat)) ; both row sums in ONE exchange.
      // The saved x-hat comes back here, after the loop (56 coalesced loads in flight at once): held across the MFMA loop it
      // would not fit beside G and the accumulators (256 VGPRs at two waves per SIMD).
      float xh[NR][2][4];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            xh[r][ct][e] = (16 * (rh + 2 * r) < TW_ROWS) ? sv.xhat[(((size_t)blockIdx.x * nl + layer) * 56 + (r * 2 + ct) * 4 + e) * 512 + tid] : 0.0f;
      const float gm0 = a.gamma[layer * BB_C + col0], gm1 = a.gamma[layer * BB_C + col0 + 16];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (rh + 2 * r) + 4 * g + e;
          const float t0 = acc[r][0][e] * gm0, t1 = acc[r][1][e] * gm1;
          acc[r][0][e] = t0; acc[r][1][e] = t1;
          const float s1 = group16_sum(t0 + t1);
          const float s2 = group16_sum(t0 * xh[r][0][e] + t1 * xh[r][1][e]);
          if (j == 0 && row < TW_ROWS) { psum[cg * TW_ROWS + row] = s1; psum[(4 + cg) * TW_ROWS + row] = s2; }
        }
      __syncthreads();
      if (tid < 2 * TW_ROWS) {
        const float* ps = psum + (tid < TW_ROWS ? 0 : 4 * TW_ROWS) + (tid < TW_ROWS ? tid : tid - TW_ROWS);
        rstat[tid] = ((ps[0] + ps[TW_ROWS]) + (ps[2 * TW_ROWS] + ps[3 * TW_ROWS])) * (1.0f / BB_C);
      }
      const float* rsd = sv.rstd + ((size_t)blockIdx.x * nl + layer) * TW_ROWS;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int rb = min(16 * (rh + 2 * r) + 4 * g, TW_ROWS - 4);
        const float4 m1 = *reinterpret_cast<const float4*>(rstat + rb);
        const float4 m2 = *reinterpret_cast<const float4*>(rstat + TW_ROWS + rb);
        const float4 rs = *reinterpret_cast<const float4*>(rsd + rb);
        const float m1v[4] = {m1.x, m1.y, m1.z, m1.w}, m2v[4] = {m2.x, m2.y, m2.z, m2.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          G[r][0][e] += rsv[e] * (acc[r][0][e] - m1v[e] - xh[r][0][e] * m2v[e]);
          G[r][1][e] += rsv[e] * (acc[r][1][e] - m1v[e] - xh[r][1][e] * m2v[e]);
        }
      }
    }
  }
  // ---- the first layer's transpose: (G * relu'_first) -> image -> dx[p][c] = sum_t sum_co W_first[co][c][t] g0[p - (t - 4)][co]
  {
    const unsigned long long mk = sv.mask[(size_t)blockIdx.x * (nl + 2) * 512 + tid];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * (rh + 2 * r) + 4 * g + e;
        if (row < TW_ROWS) {
          img[row * BB_AP + col0] = ((mk >> ((r * 2) * 4 + e)) & 1ull) && row < tile_rows ? G[r][0][e] : 0.0f;
          img[row * BB_AP + col0 + 16] = ((mk >> ((r * 2 + 1) * 4 + e)) & 1ull) && row < tile_rows ? G[r][1][e] : 0.0f;
        }
      }
  }
  __syncthreads();
  for (int e = tid; e < L * 5; e += 512) {
    const int p = e / 5, c = e - 5 * p;
    float sm = 0.0f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int q = p - (t - 4);
      if (q < 0 || q >= L) continue;
      const float4* hr = reinterpret_cast<const float4*>(img + q * BB_AP);
      const float4* wv = reinterpret_cast<const float4*>(Bs + (t * 5 + c) * BB_C);
#pragma unroll 8
      for (int k = 0; k < BB_C / 4; ++k) {
        const float4 h4 = hr[k], w4 = wv[k];
        sm += h4.x * w4.x + h4.y * w4.y + h4.z * w4.z + h4.w * w4.w;
      }
    }
    a.dx[row0 * 5 + e] = sm;
  }
}

// ---------------------------------------------------- the same forward on SEVERAL workgroups per sequence (round 4) ----
// backbone_kernel takes ~2.1 ms per workgroup whatever the batch: 4 sequences (BASELINE.json configs[0]) used 4 of 256 CUs
// for as long as 256 sequences use all of them. Here R = 2 or 4 workgroups share one sequence (104 < L <= 208), split BY ROWS:
// workgroup q owns the row tiles {2 q, 2 q + 1} + 2 R r (wave (cg, rh): tiles rh + 2 q + 2 R r), all 128 channels of them —
// so the residual stream, the LayerNorm statistics (a row's 128 channels live in one workgroup), the first-layer lookup and
// both 1 x 1 convolutions stay workgroup-local, every accumulator sees the products of backbone_kernel in the same order
// (SAME BITS, tests/test_fused_gpu.py), and the only exchange is the LayerNorm'd image: a dilated tap reads rows up to 256
// away, so each workgroup publishes its 32-row blocks of hn to a global scratch image after the LayerNorm of a layer, the R
// workgroups meet at an agent-scope counter, and each copies the blocks it does not own into its LDS image (106 KB per
// sequence and layer through the L2: the members of a group are given workgroup ids 8 apart = the same XCD). The scratch
// image is double-buffered by layer parity (a fast workgroup may publish layer l + 1 while a slow one still reads layer l).
// All R workgroups of a group must be resident: the launcher only uses this kernel when n R <= CUs.
struct BackboneSplitWs { float* xchg; int* cnt; int* err; int groups; };

template <int R>
__global__ __launch_bounds__(512, 2) void backbone_split_kernel(BackboneArgs a, BackboneSplitWs ws) {
  constexpr int TS = 2 * R;                               // stride of a wave's row tiles
  constexpr int NR = (TW_RT + TS - 1) / TS;               // R = 2: 4 ; R = 4: 2
  static_assert(NR % 2 == 0, "the two A-fragment sets keep their roles from entry to entry");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* img = smem + BB_AP;
  float* Bs = smem + (TW_ROWS + 2) * BB_AP;
  float* psum = Bs + 9 * 5 * BB_C;
  float* rstat = psum + 8 * TW_ROWS;
  int* toks = reinterpret_cast<int*>(rstat + TW_ROWS);
  int* rpos = toks + TW_ROWS;
  int* sdil = rpos + TW_ROWS;
  int* sched = sdil + BB_MAXL + 1;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = w & 3, rh = w >> 2;
  const int j = lane & 15, g = lane >> 4;
  const int col0 = 32 * cg + j;
  const int L = a.L;
  // workgroup -> (group = sequence, member q): the R members of a group sit 8 workgroup ids apart (one XCD, one L2)
  const int bid = blockIdx.x, kq = bid >> 3;
  const int q = kq % R, grp = (kq / R) * 8 + (bid & 7);
  if (grp >= a.n) return;                                 // (the whole group leaves together)
  const int t0 = rh + 2 * q;                              // this wave's row tiles: t0 + TS r
  const int tile_rows = L;
  const int64_t row0 = (int64_t)grp * L;
  const int nl = a.nl;
  const int it_end = (nl + 1) * 36;

  for (int e = tid; e < TW_ROWS; e += 512) {
    toks[e] = e < tile_rows ? (int)a.x[row0 + e] : -1;
    rpos[e] = e < tile_rows ? e : -(1 << 20);
  }
  for (int e = tid; e < BB_AP; e += 512) { smem[e] = 0.0f; img[TW_ROWS * BB_AP + e] = 0.0f; }
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < BB_MAXL; ++i) sdil[i] = a.dil[i];
    sdil[BB_MAXL] = 1;
  }
  for (int e = tid; e < 9 * 5 * BB_C; e += 512) Bs[e] = a.table0[e];
  __syncthreads();
  for (int k = tid; k < it_end; k += 512) {               // the schedule of backbone_kernel<true>
    auto entry = [&](int kk) {
      const int layer = kk / 36, t = kk % 9;
      if (layer >= nl) return t == 4 ? 0x1fff : 0;
      const int d = (t - 4) * sdil[layer];
      const int lo = d < 0 ? -d : 0, hi = d > 0 ? L - d : L;
      if (lo >= hi) return 0;
      int m = 0;
      for (int r = 0; r < TW_RT; ++r) if (lo < 16 * r + 16 && hi > 16 * r) m |= 1 << r;
      return m;
    };
    const int m = entry(k);
    int nx = k + 1;
    while (nx < it_end && entry(nx) == 0) ++nx;
    sched[k] = m ? (m | ((k % 36) / 9) << 13 | (k % 9) << 15 | nx << 19) : 0;
  }

  f32x4 f[NR][2], acc[NR][2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int col = col0 + 16 * ct;
    const float b0 = a.vec[col];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = 16 * (t0 + TS * r) + 4 * g + e;
        float v = b0;
        if (row < TW_ROWS) {
          const int pos = rpos[row];
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int pp = pos + t - 4;
            const int tk = (pp >= 0 && pp < L) ? toks[row + t - 4] : -1;
            if (tk >= 0) v += Bs[(t * 5 + tk) * BB_C + col];
          }
        }
        f[r][ct][e] = row < tile_rows ? fmaxf(v, 0.0f) : 0.0f;
      }
  }
  __syncthreads();

  const int arow0 = ((16 * t0 + j) * BB_AP + 8 * g) * 4;
  const int a_lo = arow0 - (16 * t0 + j + 1) * BB_AP * 4;
  const int a_hi = arow0 + (TW_ROWS - 16 * t0 - j) * BB_AP * 4;
  const char* imgb = reinterpret_cast<const char*>(img);
  const float* wsrc = a.tiles + col0 * CH + 8 * g;
  auto tile_of = [&](int k) { return k < nl * 36 ? k : nl * 36 + (k - nl * 36) / 9; };
  int it = 0;
  while (it < it_end && sched[it] == 0) ++it;
  it = __builtin_amdgcn_readfirstlane(it);
  int en = __builtin_amdgcn_readfirstlane(sched[it]);
  float4 bn[4];
  {
    const float* src = wsrc + (size_t)tile_of(it) * BB_C * CH;
    bn[0] = *reinterpret_cast<const float4*>(src);
    bn[1] = *reinterpret_cast<const float4*>(src + 4);
    bn[2] = *reinterpret_cast<const float4*>(src + 16 * CH);
    bn[3] = *reinterpret_cast<const float4*>(src + 16 * CH + 4);
  }
  float* const xg = ws.xchg + (size_t)grp * 2 * TW_ROWS * BB_C;      // this group's two scratch images
  int* const gcnt = ws.cnt + grp;

  for (int layer = 0; layer <= nl; ++layer) {
    const float* vl = a.vec + (size_t)(layer + 1) * 4 * BB_C;
    if (layer < nl) {
      const float tb0 = vl[BB_C + col0], tb1 = vl[BB_C + col0 + 16];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (t0 + TS * r) + 4 * g + e;
          const float sm = group16_sum((f[r][0][e] + tb0) + (f[r][1][e] + tb1));
          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sm;
        }
      __syncthreads();
      if (tid < TW_ROWS)
        rstat[tid] = ((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) * (1.0f / BB_C);
      __syncthreads();
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float4 cur4 = *reinterpret_cast<const float4*>(rstat + min(16 * (t0 + TS * r) + 4 * g, TW_ROWS - 4));
        const float mean4[4] = {cur4.x, cur4.y, cur4.z, cur4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (t0 + TS * r) + 4 * g + e;
          const float mean = mean4[e];
          const float d0 = f[r][0][e] + tb0 - mean, d1 = f[r][1][e] + tb1 - mean;
          acc[r][0][e] = d0; acc[r][1][e] = d1;
          const float sq = group16_sum(d0 * d0 + d1 * d1);
          if (j == 0 && row < TW_ROWS) psum[cg * TW_ROWS + row] = sq;
        }
      }
      __syncthreads();
      if (tid < TW_ROWS)
        rstat[tid] = rsqrtf(((psum[tid] + psum[TW_ROWS + tid]) + (psum[2 * TW_ROWS + tid] + psum[3 * TW_ROWS + tid])) *
                            (1.0f / BB_C) + 1e-5f);
      __syncthreads();
      const float gm0 = vl[2 * BB_C + col0], gm1 = vl[2 * BB_C + col0 + 16];
      const float bt0 = vl[3 * BB_C + col0], bt1 = vl[3 * BB_C + col0 + 16];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const float4 cur4 = *reinterpret_cast<const float4*>(rstat + min(16 * (t0 + TS * r) + 4 * g, TW_ROWS - 4));
        const float rs4[4] = {cur4.x, cur4.y, cur4.z, cur4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (t0 + TS * r) + 4 * g + e;
          if (row < TW_ROWS) {
            const float rs = rs4[e];
            img[row * BB_AP + col0] = row < tile_rows ? acc[r][0][e] * rs * gm0 + bt0 : 0.0f;
            img[row * BB_AP + col0 + 16] = row < tile_rows ? acc[r][1][e] * rs * gm1 + bt1 : 0.0f;
          }
        }
      }
      // ---- exchange of the LayerNorm'd image: publish the 32-row blocks this workgroup owns (blocks q, q + R, ...), meet,
      //      fetch the others'. 32 float4 per row; one row per 32 lanes.
      __syncthreads();
      float* const xb = xg + (size_t)(layer & 1) * TW_ROWS * BB_C;
      // The scratch image is written and read with SYSTEM-SCOPE accesses (sc0 sc1: write-through to memory, reads past every
      // cache) and the counter with relaxed agent-scope atomics. The first version used ordinary stores / loads between
      // __threadfence()s: an agent-scope release / acquire on this part writes back and invalidates the XCD's whole L2 — the
      // 13 MB of weights every workgroup streams from it went with it, 25 us per layer (B = 64: slower than one workgroup per
      // sequence). Nothing but these 106 KB per layer needs to be coherent between the workgroups.
      for (int e = tid; e < NR * 32 * 32; e += 512) {
        const int c4 = e & 31, rr = e >> 5;
        const int row = 32 * (q + R * (rr >> 5)) + (rr & 31);
        if (row < TW_ROWS) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(img + row * BB_AP + 4 * c4);
          asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(xb + (size_t)row * BB_C + 4 * c4), "v"(v) : "memory");
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this thread's blocks have reached memory
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_fetch_add(gcnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int target = R * (layer + 1);
        int spins = 0;
        // (once a barrier of this launch has timed out the launch's results are void — svdd_backbone_split_status reports it, the
        //  samplers raise — and the remaining layers do not wait again)
        while (__hip_atomic_load(gcnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
          __builtin_amdgcn_s_sleep(1);
          if ((++spins & 0xFFF) == 0 && __hip_atomic_load(ws.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
          if (spins > (1 << 24)) { __hip_atomic_store(ws.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }  // a member is not resident: give up, never hang
        }
      }
      __syncthreads();
      // every foreign block of the image in ONE batch of loads per lane (14 slots of 512 lanes x 16 B cover the 208 x 512 B
      // image; the slots of the blocks this workgroup owns stay empty): one memory latency per layer, not one per four loads
      {
        f32x4 v[14];
        bool ok[14];
#define S_XLD(K)                                                                                             \
        { const int e_ = (K) * 512 + tid; const int row_ = e_ >> 5;                                          \
          ok[K] = row_ < TW_ROWS && ((row_ >> 5) % R) != q;                                                  \
          v[K] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};                                                              \
          if (ok[K]) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[K]) : "v"(xb + (size_t)row_ * BB_C + 4 * (e_ & 31)) : "memory"); }
        S_XLD(0) S_XLD(1) S_XLD(2) S_XLD(3) S_XLD(4) S_XLD(5) S_XLD(6) S_XLD(7) S_XLD(8) S_XLD(9) S_XLD(10) S_XLD(11) S_XLD(12) S_XLD(13)
#undef S_XLD
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                     "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]) :: "memory");
#pragma unroll
        for (int k = 0; k < 14; ++k) {
          const int e_ = k * 512 + tid;
          if (ok[k]) *reinterpret_cast<f32x4*>(img + (e_ >> 5) * BB_AP + 4 * (e_ & 31)) = v[k];
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (t0 + TS * r) + 4 * g + e;
          if (row < TW_ROWS) { img[row * BB_AP + col0] = f[r][0][e]; img[row * BB_AP + col0 + 16] = f[r][1][e]; }
        }
    }
    const float bl0 = vl[col0], bl1 = vl[col0 + 16];
#pragma unroll
    for (int r = 0; r < NR; ++r) { acc[r][0] = f32x4{bl0, bl0, bl0, bl0}; acc[r][1] = f32x4{bl1, bl1, bl1, bl1}; }
    const int dil = __builtin_amdgcn_readfirstlane(sdil[layer < nl ? layer : BB_MAXL]);
    const int layer_end = (layer + 1) * 36;
    __syncthreads();                                      // the image is complete
#define S_ALOAD(R_, V, COFF, DBYTES)                                                                         \
      { const int o_ = min(max(arow0 + (DBYTES) + (R_) * (16 * TS * BB_AP * 4), a_lo + (COFF)), a_hi + (COFF)); \
        const float4* ap_ = reinterpret_cast<const float4*>(imgb + o_);                                      \
        V[0] = ap_[0]; V[1] = ap_[1]; }
#define S_MM(R_, U)                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
      __builtin_amdgcn_s_waitcnt(0xC07F | (2 << 8));      /* at least two LDS reads (the other set's) were issued after U's */ \
      if (live & (1 << (TS * (R_)))) {                                                                       \
        _Pragma("unroll") for (int qq = 0; qq < 2; ++qq) {                                                   \
          acc[R_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].x, bf0[4 * qq], acc[R_][0], 0, 0, 0);      \
          acc[R_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].x, bf1[4 * qq], acc[R_][1], 0, 0, 0);      \
          acc[R_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].y, bf0[4 * qq + 1], acc[R_][0], 0, 0, 0);  \
          acc[R_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].y, bf1[4 * qq + 1], acc[R_][1], 0, 0, 0);  \
          acc[R_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].z, bf0[4 * qq + 2], acc[R_][0], 0, 0, 0);  \
          acc[R_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].z, bf1[4 * qq + 2], acc[R_][1], 0, 0, 0);  \
          acc[R_][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].w, bf0[4 * qq + 3], acc[R_][0], 0, 0, 0);  \
          acc[R_][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(U[qq].w, bf1[4 * qq + 3], acc[R_][1], 0, 0, 0);  \
        }                                                                                                    \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);
#define S_PARAMS(EN, COFF, DBYTES)                                                                           \
      const int COFF = (((EN) >> 13) & 3) * (CH * 4);                                                        \
      const int DBYTES = ((((EN) >> 15) & 15) - 4) * dil * (BB_AP * 4) + COFF;
    float4 ua[2], ub[2];
    if (it < layer_end) {
      S_PARAMS(en, coff0, dbytes0)
      S_ALOAD(0, ua, coff0, dbytes0)
      S_ALOAD(1, ub, coff0, dbytes0)
    }
    while (it < layer_end) {
      const int nxt = en >> 19;
      const int en_next_v = sched[nxt < it_end ? nxt : it];
      const float bf0[8] = {bn[0].x, bn[0].y, bn[0].z, bn[0].w, bn[1].x, bn[1].y, bn[1].z, bn[1].w};
      const float bf1[8] = {bn[2].x, bn[2].y, bn[2].z, bn[2].w, bn[3].x, bn[3].y, bn[3].z, bn[3].w};
      if (nxt < it_end) {
        const float* src = wsrc + (size_t)tile_of(nxt) * BB_C * CH;
        bn[0] = *reinterpret_cast<const float4*>(src);
        bn[1] = *reinterpret_cast<const float4*>(src + 4);
        bn[2] = *reinterpret_cast<const float4*>(src + 16 * CH);
        bn[3] = *reinterpret_cast<const float4*>(src + 16 * CH + 4);
      }
      S_PARAMS(en, coff, dbytes)
      const int live = (en & 0x1fff) >> t0;               // bit TS r = owned tile r (tiles >= 13 do not exist)
      S_MM(0, ua)
      const int en2 = __builtin_amdgcn_readfirstlane(en_next_v);
      S_PARAMS(en2, coff2, dbytes2)
      if constexpr (NR == 2) {
        S_ALOAD(0, ua, coff2, dbytes2)
        S_MM(1, ub)
        S_ALOAD(1, ub, coff2, dbytes2)
      } else {
        S_ALOAD(2, ua, coff, dbytes)
        S_MM(1, ub)
        S_ALOAD(3, ub, coff, dbytes)
        S_MM(2, ua)
        S_ALOAD(0, ua, coff2, dbytes2)
        S_MM(3, ub)
        S_ALOAD(1, ub, coff2, dbytes2)
      }
      it = nxt;
      en = en2;
    }
#undef S_PARAMS
#undef S_MM
#undef S_ALOAD
    __syncthreads();
    if (layer < nl) {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int e = 0; e < 4; ++e) f[r][ct][e] = fmaxf(acc[r][ct][e], 0.0f) + f[r][ct][e];
    } else {
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * (t0 + TS * r) + 4 * g + e;
          if (row < TW_ROWS) {
            img[row * BB_AP + col0] = fmaxf(acc[r][0][e], 0.0f);
            img[row * BB_AP + col0 + 16] = fmaxf(acc[r][1][e], 0.0f);
          }
        }
    }
  }
  __syncthreads();
  // ---- last 1x1 conv 128 -> 5 on the rows this workgroup owns
  for (int e = tid; e < NR * 32 * 5; e += 512) {
    const int rr = e / 5, v = e - 5 * rr;
    const int row = 32 * (q + R * (rr >> 5)) + (rr & 31);
    if (row >= tile_rows) continue;
    const float* hr = img + row * BB_AP;
    const float* wv = a.w2 + v * BB_C;
    float sm = a.w2[5 * BB_C + v];
#pragma unroll 8
    for (int k = 0; k < BB_C; ++k) sm += hr[k] * wv[k];
    a.out[(row0 + row) * 5 + v] = sm;
  }
}

// ------------------------------------------------------- value-net tail: everything after the GRU in one pass ----
// score[n][t] = b_eff[t] + mean_l sum_c w_eff[c][t] * relu(b1[c] + sum_k W1[c][k] * LayerNorm(h_fwd + h_bwd)[n][l][k])
// (reference Enformer.py:1617 direction sum, :2010-2047 FeedForwardBlock LayerNorm -> Linear 64->128 -> ReLU ->
//  Linear 128->64, :2131-2173 ConvHead 1x1 conv + mean over length; the last two linear maps are collapsed and the
//  LayerNorm affine is folded into (W1, b1) by the host).
// Replaces epilogue_ln + 2 GEMMs + ReLU + mean: those moved 1.3 GB through HBM per value forward (n = 2560, L = 200);
// this reads the two GRU outputs once (262 MB) and writes n*T floats. One wave per sequence; 16-row tiles; the
// 64 -> 128 map runs on the exact-fp32 matrix cores with W1 held in registers as MFMA B operands; LayerNorm statistics
// are two-pass across the 4 lanes of a row. The next tile's rows are loaded RAW into registers under the current
// tile's MFMAs and only summed when their turn comes (summing at load time would park the wave on the loads).
template <int T>
__global__ __launch_bounds__(256, 2) void value_tail_kernel(const float* __restrict__ hf, const float* __restrict__ hb,
                                                            const float* __restrict__ w1pack, const float* __restrict__ b1,
                                                            const float* __restrict__ weff, const float* __restrict__ beff,
                                                            float* __restrict__ out, int n_alloc, int L,
                                                            const int* __restrict__ count) {
  // b1' and w_eff per output column, [j][8 ct | 8 ct x T]: read from LDS every tile (lgkmcnt, not the row prefetch's vmcnt) instead of
  // living in 8 + 8 T registers — with W1' (128), the accumulators (32) and two row sets the kernel sat at the 256-register limit and
  // spilled its lane offsets, reloading them from scratch every tile.
  __shared__ __attribute__((aligned(16))) float colv[16][8 + 8 * T];
  for (int e = threadIdx.x; e < 16 * (8 + 8 * T); e += 256) {
    const int jj = e / (8 + 8 * T), r = e - jj * (8 + 8 * T);
    colv[jj][r] = r < 8 ? b1[16 * r + jj] : weff[(16 * ((r - 8) / T) + jj) * T + (r - 8) % T];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int seq = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));   // wave-uniform: the row bases stay in SGPRs
  const int n = count ? *count : n_alloc;
  if (seq >= n) return;
  const int j = lane & 15, g = lane >> 4;
  // channel of this lane's value i (= MFMA k-step i): ch(i) = 16 (i / 4) + 4 g + (i % 4) — four 16-byte pieces that,
  // across the 4 lanes of a row, make each load instruction cover whole 64-byte segments
  float wb[128];                                           // wb[16 ct + s] = W1'[16 ct + j][ch(s)]
  {
    const float4* wp = reinterpret_cast<const float4*>(w1pack + (size_t)lane * 128);
#pragma unroll
    for (int i = 0; i < 32; ++i) { const float4 v = wp[i]; wb[4 * i] = v.x; wb[4 * i + 1] = v.y; wb[4 * i + 2] = v.z; wb[4 * i + 3] = v.w; }
  }
  // Round 6: the operands above are PINNED here (they must have arrived before the loop). Left pending, their loads share the
  // in-order vmcnt counter with the row prefetches, and the compiler's conservative waits inside the loop (s_waitcnt vmcnt(3..1) in
  // the middle of the MFMA block) drained each tile's prefetch a third of the way into the tile it was meant to fly under.
#pragma unroll
  for (int i = 0; i < 128; ++i) asm volatile("" : "+v"(wb[i]));
  float part[T];
#pragma unroll
  for (int t = 0; t < T; ++t) part[t] = 0.0f;
  const float* sf = hf + (size_t)seq * L * 64;             // scalar bases + a 32-bit lane offset (L * 64 floats < 2^31 bytes per sequence)
  const float* sb = hb + (size_t)seq * L * 64;
  const int ntiles = (L + 15) / 16;
  float4 xa[4], xb[4];
  auto load_rows = [&](int tile) {
    const int off = min(16 * tile + j, L - 1) * 64 + 4 * g;   // rows past the end re-read the last row; masked below
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      xa[i] = *reinterpret_cast<const float4*>(sf + off + 16 * i);
      xb[i] = *reinterpret_cast<const float4*>(sb + off + 16 * i);
    }
  };
  load_rows(0);
  for (int tile = 0; tile < ntiles; ++tile) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[4 * i] = xa[i].x + xb[i].x; v[4 * i + 1] = xa[i].y + xb[i].y;
      v[4 * i + 2] = xa[i].z + xb[i].z; v[4 * i + 3] = xa[i].w + xb[i].w;
    }
    // the next tile's loads go into the registers the sums above just freed (hoisted above the sums, as the scheduler would, they
    // need a second set of 32 registers + 32 copies per tile, and that spilled the row pointers into scratch inside the loop)
    __builtin_amdgcn_sched_barrier(0);
    load_rows(min(tile + 1, ntiles - 1));                 // fly under the LayerNorm and the 128 MFMAs below (unconditional: a branch
    __builtin_amdgcn_sched_barrier(0);                    // here becomes a block of its own, placed BEFORE the sums; the last tile re-reads itself)
    // LayerNorm (no affine: folded into W1', b1') over the 64 channels of row 16 tile + j: 16 here, 48 in lanes j + 16 g'
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
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] *= rstd;
    int jj = j;
    asm volatile("" : "+v"(jj));                           // opaque per tile: the LDS reads stay in the loop (the INDEX, not the pointer:
    const float* cv = colv[jj];                            // a laundered pointer turns generic and its flat loads wait on vmcnt too)
    f32x4 acc[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) { const float b = cv[ct]; acc[ct] = f32x4{b, b, b, b}; }
#pragma unroll
    for (int sidx = 0; sidx < 16; ++sidx)
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[sidx], wb[16 * ct + sidx], acc[ct], 0, 0, 0);
    // C/D layout: reg rho -> row 4 g + rho, column 16 ct + j
    constexpr bool HOIST = T <= 3;                         // w_eff of this lane's columns: one LDS read per tile, not one per row
    float wl[HOIST ? 8 * T : 1];                           // (T = 4: 32 more registers would spill; read per row there)
    if constexpr (HOIST) {
#pragma unroll
      for (int i = 0; i < 8 * T; ++i) wl[i] = cv[8 + i];
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      if (16 * tile + 4 * g + rho < L) {
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          const float z = fmaxf(acc[ct][rho], 0.0f);
#pragma unroll
          for (int t = 0; t < T; ++t) part[t] += z * (HOIST ? wl[ct * T + t] : cv[8 + ct * T + t]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const float tot = wave_sum(part[t]);
    if (lane == 0) out[(size_t)seq * T + t] = tot / (float)L + beff[t];
  }
}

}  // namespace

static int g_fixed_spt = 0;        // svdd_set_backbone_packing: 1 = always fill the 208-row tile (round-1 behaviour), -s = exactly
                                   // s sequences per tile (calibration of svdd_spt.h), 0 = choose (default)
extern "C" int svdd_set_backbone_packing(int full) { g_fixed_spt = full; return SVDD_OK; }
extern "C" int svdd_internal_fixed_spt() { return g_fixed_spt; }
extern "C" int svdd_internal_num_cus() {
  static int ncu = 0;
  if (!ncu) {
    hipDeviceProp_t prop;
    int dev = 0;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  return ncu;
}

static int g_gru_mode = 0;         // 0 / 4: producer-consumer kernel (default) ; 1: one wave does both halves ; 2: both
                                   // directions per workgroup ; 3: mode 1 with an LDS reservation (experiments, A/B tests)
extern "C" int svdd_gru_set_mode(int mode) { g_gru_mode = mode; return SVDD_OK; }

extern "C" int svdd_gru_bidir_f32(const float* x, const float* wpack, const float* bpack, float* out, int n, int L,
                                  const int32_t* count, void* stream) {
  if (!x || !wpack || !bpack || !out || n <= 0 || L <= 0) return SVDD_E_ARG;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(3, &e0, &e1);
  // both directions per workgroup when that fills the chip evenly (<= 16 sequences per CU); else one direction each
  const int per_cu = (n + 255) / 256;
  if (g_gru_mode == 0 || g_gru_mode == 4) {          // default: producer / consumer waves (same bits as mode 1)
    hipExtLaunchKernelGGL(gru_pc_kernel, dim3(2 * (unsigned)((n + TS - 1) / TS)), dim3(512), 0, (hipStream_t)stream, e0, e1, 0,
                          x, wpack, bpack, out, n, L, count);
    return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
  }
  if (g_gru_mode == 2 && per_cu <= TS && n >= 256)   // measured slower (841 vs 758 us at n=2560, L=200): opt-in only
    hipExtLaunchKernelGGL(gru_bidir_kernel<true>, dim3((unsigned)((n + per_cu - 1) / per_cu)), dim3(512), 0,
                          (hipStream_t)stream, e0, e1, 0, x, wpack, bpack, out, n, L, per_cu, count);
  else {
    // (Reserving > 80 KB of dynamic LDS to force one workgroup per CU was measured: 2560 sequences = 320 units take
    //  749 us in two exclusive rounds against 678 us shared, and <= 256 units spread over the CUs by themselves: 367 us
    //  either way. Kept as mode 3 for the record.)
    const size_t pad = g_gru_mode == 3 ? 84 * 1024 : 0;
    if (pad) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_bidir_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
    hipExtLaunchKernelGGL(gru_bidir_kernel<false>, dim3(2 * (unsigned)((n + TS - 1) / TS)), dim3(256), pad,
                          (hipStream_t)stream, e0, e1, 0, x, wpack, bpack, out, n, L, TS, count);
  }
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_epilogue_ln_f32(const float* y, const float* bias, const float* f_prev, const float* tb,
                                    const float* gamma, const float* beta, float* f_out, float* hn, int64_t rows,
                                    int channels, int act, void* stream) {
  if (!y || (!f_out && !hn) || rows <= 0 || (hn && (!gamma || !beta)) || act < 0 || act > 2) return SVDD_E_ARG;
  EpiArgs a{y, bias, f_prev, tb, gamma, beta, f_out, hn, rows, act};
  const int64_t nblocks = (rows + 3) / 4;
  const unsigned grid = (unsigned)(nblocks < 4096 ? nblocks : 4096);
  hipEvent_t e0, e1;
  if (channels != 64 && channels != 128 && channels != 256) return SVDD_E_ARG;
  svdd_internal_timed_events(4, &e0, &e1);
  if (channels == 64) hipExtLaunchKernelGGL(epilogue_ln_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  else if (channels == 128) hipExtLaunchKernelGGL(epilogue_ln_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  else hipExtLaunchKernelGGL(epilogue_ln_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, e0, e1, 0, a);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

static int g_tower_ver = 0;        // svdd_set_tower_version: 0 = default (= 3), 1 = first generation, 2 / 3 = second generation
extern "C" int svdd_set_tower_version(int v) { g_tower_ver = v; return SVDD_OK; }   // with 2 / 1 column tiles per wave

static int g_conv_dynamic = 0;     // tests: force the dynamically scheduled kernel
extern "C" int svdd_conv1d_set_dynamic(int on) { g_conv_dynamic = on; return SVDD_OK; }

extern "C" int svdd_conv1d_cl_f32(const float* x, const float* wpack, float* y, int n, int L, int cin, int cout,
                                  int taps, int dilation, const float* bias, const float* f_prev, int act,
                                  const float* tb, const float* gamma, const float* beta, float* hn, void* stream) {
  if (!x || !wpack || !y || n <= 0 || L <= 0 || L > CONV_ROWS || taps <= 0 || !(taps & 1) || dilation <= 0 ||
      act < -1 || act > 2 || (hn && (act < 0 || !gamma || !beta)))
    return SVDD_E_ARG;
  const int spt = CONV_ROWS / L;                               // whole sequences per workgroup tile
  ConvArgs a{x, wpack, y, n, L, spt, taps, dilation, bias, f_prev, act, tb, gamma, beta, hn};
  const unsigned grid = (unsigned)((n + spt - 1) / spt);
  auto launch = [&](auto kern, int co) {
    const size_t lds = sizeof(float) * ((size_t)(CONV_ROWS + 1) * CHP + 2 * (size_t)co * CHP);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    svdd_internal_timed_events(2, &e0, &e1);
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, (hipStream_t)stream, e0, e1, 0, a);
  };
  const bool dyn = g_conv_dynamic != 0;
  if (!dyn && cin == 128 && cout == 128 && taps == 9 && (L == 200 || L == 50)) {
#define SVDD_CONV_CASE(D, LL) if (dilation == D && L == LL) { launch(conv1d_cl_static_kernel<128, 128, 9, D, LL>, 128); return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH; }
    SVDD_CONV_CASE(1, 200) SVDD_CONV_CASE(4, 200) SVDD_CONV_CASE(16, 200) SVDD_CONV_CASE(64, 200)
    SVDD_CONV_CASE(1, 50) SVDD_CONV_CASE(4, 50) SVDD_CONV_CASE(16, 50) SVDD_CONV_CASE(64, 50)
#undef SVDD_CONV_CASE
  }
  if (!dyn && cin == 64 && cout == 64 && taps == 5 && dilation == 1 && (L == 200 || L == 50)) {
    if (L == 200) launch(conv1d_cl_static_kernel<64, 64, 5, 1, 200>, 64);
    else launch(conv1d_cl_static_kernel<64, 64, 5, 1, 50>, 64);
    return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
  }
  if (act >= 0) return SVDD_E_ARG;                             // the generic kernel has no fused epilogue
  if (cin == 128 && cout == 128) launch(conv1d_cl_kernel<128, 128>, 128);
  else if (cin == 64 && cout == 64) launch(conv1d_cl_kernel<64, 64>, 64);
  else if (cin == 64 && cout == 128) launch(conv1d_cl_kernel<64, 128>, 128);
  else if (cin == 128 && cout == 64) launch(conv1d_cl_kernel<128, 64>, 64);
  else return SVDD_E_ARG;
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

// y = gate > 0 ? conv(x) + f_prev : 0 — one layer of the reward tower's BACKWARD pass (DPS): x = the gradient at this layer's
// pre-activation, wpack = the transposed / flipped taps, f_prev = the residual branch's gradient (NULL: none), gate = the forward
// activation of the layer BELOW (its ReLU decides where the gradient passes; NULL: no gating). 64 -> 64 x 5 taps, L = 200 / 50 only.
extern "C" int svdd_conv1d_cl_gated_f32(const float* x, const float* wpack, float* y, int n, int L, int cin, int cout, int taps,
                                        int dilation, const float* f_prev, const float* gate, void* stream) {
  if (!x || !wpack || !y || n <= 0 || cin != 64 || cout != 64 || taps != 5 || dilation != 1 || (L != 200 && L != 50)) return SVDD_E_ARG;
  const int spt = CONV_ROWS / L;
  ConvArgs a{x, wpack, y, n, L, spt, taps, dilation, nullptr, f_prev, gate ? 3 : 2, gate, nullptr, nullptr, nullptr};
  const unsigned grid = (unsigned)((n + spt - 1) / spt);
  const size_t lds = sizeof(float) * ((size_t)(CONV_ROWS + 1) * CHP + 2 * (size_t)64 * CHP);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(2, &e0, &e1);
  if (L == 200) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_cl_static_kernel<64, 64, 5, 1, 200>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipExtLaunchKernelGGL((conv1d_cl_static_kernel<64, 64, 5, 1, 200>), dim3(grid), dim3(256), lds, (hipStream_t)stream, e0, e1, 0, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_cl_static_kernel<64, 64, 5, 1, 50>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipExtLaunchKernelGGL((conv1d_cl_static_kernel<64, 64, 5, 1, 50>), dim3(grid), dim3(256), lds, (hipStream_t)stream, e0, e1, 0, a);
  }
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_conv_tower_f32(const float* onehot, const float* tiles, const float* bias, float* out, int n, int L,
                                   int nlayers, int residual_mask, const int32_t* count, void* stream) {
  if (!onehot || !tiles || !bias || !out || n <= 0 || L <= 0 || L > TW_ROWS || nlayers <= 0 || nlayers > TW_MAXL)
    return SVDD_E_ARG;
  const int spt = TW_ROWS / L;
  TowerArgs a{onehot, tiles, bias, out, n, L, spt, nlayers, residual_mask, count};
  const size_t lds = sizeof(float) * ((size_t)(TW_ROWS + 2) * TW_AP + (size_t)(TW_ROWS + 2) * 4);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(5, &e0, &e1);
  const dim3 grid((unsigned)((n + spt - 1) / spt));
  if (g_tower_ver != 1) {
    TowerWinArgs wa{a, nullptr, nullptr, 1, nullptr, nullptr};
    const size_t lds2 = sizeof(float) * ((size_t)(TW_ROWS + 2) * TW_AP + (size_t)(TW_ROWS + 16) * 4);
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipExtLaunchKernelGGL(kern, grid, dim3(512), lds2, (hipStream_t)stream, e0, e1, 0, wa);
    };
    if (g_tower_ver == 2) { if (spt == 1) go(conv_tower2_kernel<true, false, 2>); else go(conv_tower2_kernel<false, false, 2>); }
    else { if (spt == 1) go(conv_tower2_kernel<true, false, 1>); else go(conv_tower2_kernel<false, false, 1>); }
    return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
  }
  if (spt == 1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tower_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipExtLaunchKernelGGL(conv_tower_kernel<true>, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tower_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipExtLaunchKernelGGL(conv_tower_kernel<false>, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a);
  }
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

// Scratch of backbone_split_kernel, owned by the caller (the library never allocates): `bytes` of device memory =
// groups * (2 * 208 * 128 floats) for the double-buffered LayerNorm images + groups + 1 ints (arrival counters, error flag).
// One workspace per process: launches that use it must not overlap (the engine issues its backbone forwards on one stream).
static BackboneSplitWs g_bb_ws = {nullptr, nullptr, nullptr, 0};
static int g_bb_split = 0;            // svdd_set_option(SVDD_OPT_BACKBONE_SPLIT): 0 auto, 1 off, 2 / 4 force R where it fits
extern "C" void svdd_internal_set_bb_split(int v) { g_bb_split = (v == 1 || v == 2 || v == 4) ? v : 0; }
extern "C" int svdd_backbone_set_workspace(void* ws, long long bytes) {
  const long long per = (long long)2 * TW_ROWS * BB_C * sizeof(float) + sizeof(int);
  if (!ws || bytes < per + (long long)sizeof(int)) { g_bb_ws = {nullptr, nullptr, nullptr, 0}; return ws ? SVDD_E_ARG : SVDD_OK; }
  const int groups = (int)((bytes - (long long)sizeof(int)) / per);
  g_bb_ws.xchg = reinterpret_cast<float*>(ws);
  g_bb_ws.cnt = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + (size_t)groups * 2 * TW_ROWS * BB_C * sizeof(float));
  g_bb_ws.err = g_bb_ws.cnt + groups;
  g_bb_ws.groups = groups;
  return hipMemset(g_bb_ws.err, 0, sizeof(int)) == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}
// 0 = every group barrier of every split launch so far was met; 1 = one timed out (results of that launch are invalid). Synchronises.
extern "C" int svdd_backbone_split_status(int* err_out) {
  if (!err_out) return SVDD_E_ARG;
  *err_out = 0;
  if (!g_bb_ws.err) return SVDD_OK;
  // the split launches may sit on ANY stream (the caller passes its own): a blocking hipMemcpy orders only against the null stream
  // and blocking streams, so a launch on a non-blocking stream could still be running — and its time-out would surface one decode late
  if (hipDeviceSynchronize() != hipSuccess) return SVDD_E_LAUNCH;
  if (hipMemcpy(err_out, g_bb_ws.err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return SVDD_E_LAUNCH;
  if (*err_out && hipMemset(g_bb_ws.err, 0, sizeof(int)) != hipSuccess) return SVDD_E_LAUNCH;   // reported once, then cleared
  return SVDD_OK;
}

// DPS: the forward that also saves what the gradient kernel needs, and the gradient kernel (one launch each way)
extern "C" int svdd_backbone_cnn_save_f32(const uint8_t* x, const float* table0, const float* tiles, const float* vec,
                                          const float* w2, float* out, int n, int L, int nlayers, const int* dilations,
                                          float* xhat, float* rstd, unsigned long long* mask, void* stream) {
  if (!x || !table0 || !tiles || !vec || !w2 || !out || !dilations || !xhat || !rstd || !mask || n <= 0 || L <= TW_ROWS / 2 ||
      L > TW_ROWS || nlayers <= 0 || nlayers > BB_MAXL)
    return SVDD_E_ARG;
  BackboneArgs a;
  a.x = x; a.table0 = table0; a.tiles = tiles; a.vec = vec; a.w2 = w2; a.out = out;
  a.n = n; a.L = L; a.spt = 1; a.nl = nlayers; a.count = nullptr; a.row_idx = nullptr; a.out_scatter = 0;
  a.auto_spt = 0; a.ncu = svdd_internal_num_cus(); a.plan = SvddTilePlan{1, 0, 1};
  for (int i = 0; i < BB_MAXL; ++i) a.dil[i] = i < nlayers ? dilations[i] : 1;
  for (int i = 0; i < nlayers; ++i) if (dilations[i] <= 0) return SVDD_E_ARG;
  const size_t lds = sizeof(float) * ((size_t)(TW_ROWS + 2) * BB_AP + 9 * 5 * (size_t)BB_C + 8 * (size_t)TW_ROWS +
                                      3 * (size_t)TW_ROWS + BB_MAXL + 1 + (size_t)(nlayers + 1) * 36);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(6, &e0, &e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipExtLaunchKernelGGL((backbone_kernel<true, true>), dim3((unsigned)n), dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a,
                        BackboneSave{xhat, rstd, mask});
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_backbone_cnn_grad_f32(const float* dlogits, const float* tiles_bwd, const float* gamma, const float* w2,
                                          const float* table0, const float* xhat, const float* rstd, const unsigned long long* mask,
                                          float* dx, int n, int L, int nlayers, const int* dilations, void* stream) {
  if (!dlogits || !tiles_bwd || !gamma || !w2 || !table0 || !xhat || !rstd || !mask || !dx || !dilations || n <= 0 ||
      L <= TW_ROWS / 2 || L > TW_ROWS || nlayers <= 0 || nlayers > BB_MAXL)
    return SVDD_E_ARG;
  BackboneGradArgs a;
  a.dlogits = dlogits; a.tiles = tiles_bwd; a.gamma = gamma; a.w2 = w2; a.table0 = table0; a.dx = dx;
  a.n = n; a.L = L; a.nl = nlayers;
  for (int i = 0; i < BB_MAXL; ++i) a.dil[i] = i < nlayers ? dilations[i] : 1;
  for (int i = 0; i < nlayers; ++i) if (dilations[i] <= 0) return SVDD_E_ARG;
  const size_t lds = sizeof(float) * ((size_t)(TW_ROWS + 2) * BB_AP + 9 * 5 * (size_t)BB_C + 8 * (size_t)TW_ROWS +
                                      2 * (size_t)TW_ROWS + BB_MAXL + 1 + (size_t)(nlayers + 1) * 36);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(10, &e0, &e1);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipExtLaunchKernelGGL(backbone_grad_kernel, dim3((unsigned)n), dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a,
                        BackboneSave{const_cast<float*>(xhat), const_cast<float*>(rstd), const_cast<unsigned long long*>(mask)});
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_backbone_cnn_f32(const uint8_t* x, const float* table0, const float* tiles, const float* vec,
                                     const float* w2, float* out, int n, int L, int nlayers, const int* dilations,
                                     const int32_t* count, const int32_t* row_idx, int out_scatter, void* stream) {
  if (!x || !table0 || !tiles || !vec || !w2 || !out || !dilations || n <= 0 || L <= 0 || L > TW_ROWS ||
      nlayers <= 0 || nlayers > BB_MAXL)
    return SVDD_E_ARG;
  BackboneArgs a;
  a.x = x; a.table0 = table0; a.tiles = tiles; a.vec = vec; a.w2 = w2; a.out = out;
  a.n = n; a.L = L; a.spt = TW_ROWS / L; a.nl = nlayers; a.count = count; a.row_idx = row_idx; a.out_scatter = out_scatter;
  a.auto_spt = 0; a.ncu = svdd_internal_num_cus();
  unsigned nwg = (unsigned)((n + a.spt - 1) / a.spt);
  a.plan = SvddTilePlan{a.spt, 0, a.spt};                // every tile full (one sequence per tile: never read)
  if (a.spt > 1 && g_fixed_spt <= 0) {                   // several sequences fit a tile: which tile takes how many (svdd_spt.h)
    if (g_fixed_spt < 0) { a.spt = -g_fixed_spt < a.spt ? -g_fixed_spt : a.spt; a.plan = SvddTilePlan{a.spt, 0, a.spt}; nwg = (unsigned)((n + a.spt - 1) / a.spt); }
    else if (count) { a.auto_spt = 1; nwg = (unsigned)n; }   // decided on the device from *count; grid for one sequence per tile
    else { a.plan = svdd_plan_tiles(n, L, a.ncu, 9); a.spt = a.plan.s2; nwg = (unsigned)svdd_plan_num_tiles(a.plan, n); }
  }
  for (int i = 0; i < BB_MAXL; ++i) a.dil[i] = i < nlayers ? dilations[i] : 1;
  for (int i = 0; i < nlayers; ++i) if (dilations[i] <= 0) return SVDD_E_ARG;
  const size_t lds = sizeof(float) * ((size_t)(TW_ROWS + 2) * BB_AP + 9 * 5 * (size_t)BB_C + 8 * (size_t)TW_ROWS +
                                      3 * (size_t)TW_ROWS + BB_MAXL + 1 + (size_t)(nlayers + 1) * 36);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(6, &e0, &e1);
  // Small batches of one-sequence tiles: R = 2 / 4 workgroups per sequence (backbone_split_kernel; same bits). Needs the scratch
  // image of svdd_backbone_set_workspace, a row count known on the host (no device-side count / index list) and n R <= CUs
  // (every member of a group must be resident: they wait for each other).
  if (a.spt == 1 && !a.auto_spt && TW_ROWS / L == 1 && !count && !row_idx && g_bb_ws.xchg && g_bb_split != 1) {
    // ... and the TAIL ROUND of a batch that is not a multiple of the CU count (257 .. 384 sequences took two full rounds of
    // one-workgroup tiles): whole rounds on backbone_kernel, the remainder — when it is at most half a round — split.
    const int rem = n % a.ncu, n_main = n - rem;
    const int ns = n_main == 0 ? n : rem;                 // sequences that go to the split kernel
    int R = g_bb_split == 2 || g_bb_split == 4 ? g_bb_split : (4 * ns <= a.ncu ? 4 : 2 * ns <= a.ncu ? 2 : 1);
    if (ns > 0 && R > 1 && (int64_t)R * ns <= a.ncu && ns <= g_bb_ws.groups) {
      if (n_main > 0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        BackboneArgs am = a;
        am.n = n_main;
        hipExtLaunchKernelGGL(backbone_kernel<true>, dim3((unsigned)n_main), dim3(512), lds, (hipStream_t)stream, e0, nullptr, 0, am, BackboneSave{});
        if (hipGetLastError() != hipSuccess) return SVDD_E_LAUNCH;
        a.x += (size_t)n_main * L; a.out += (size_t)n_main * L * 5; a.n = ns;
        e0 = nullptr;                                     // one timed span over both launches
      }
      if (hipMemsetAsync(g_bb_ws.cnt, 0, sizeof(int) * (size_t)ns, (hipStream_t)stream) != hipSuccess) return SVDD_E_LAUNCH;
      const dim3 sgrid((unsigned)(((ns + 7) / 8) * 8 * R));
      if (R == 2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_split_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipExtLaunchKernelGGL(backbone_split_kernel<2>, sgrid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a, g_bb_ws);
      } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_split_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipExtLaunchKernelGGL(backbone_split_kernel<4>, sgrid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a, g_bb_ws);
      }
      return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
    }
  }
  const dim3 grid(nwg);
  if (!a.auto_spt && a.plan.s2 == 1 && (a.plan.n1 == 0 || a.plan.s1 == 1)) {   // every tile holds one sequence
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipExtLaunchKernelGGL(backbone_kernel<true>, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a, BackboneSave{});
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(backbone_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipExtLaunchKernelGGL(backbone_kernel<false>, grid, dim3(512), lds, (hipStream_t)stream, e0, e1, 0, a, BackboneSave{});
  }
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_value_tail_f32(const float* h_fwd, const float* h_bwd, const float* w1pack, const float* b1,
                                   const float* w_eff, const float* b_eff, float* out, int n, int L, int n_tasks,
                                   const int32_t* count, void* stream) {
  if (!h_fwd || !h_bwd || !w1pack || !b1 || !w_eff || !b_eff || !out || n <= 0 || L <= 0 || n_tasks < 1 || n_tasks > 4)
    return SVDD_E_ARG;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(7, &e0, &e1);
  const dim3 grid((unsigned)((n + 3) / 4)), block(256);
#define SVDD_TAIL(TT)                                                                                              \
  hipExtLaunchKernelGGL(value_tail_kernel<TT>, grid, block, 0, (hipStream_t)stream, e0, e1, 0, h_fwd, h_bwd, w1pack, b1, \
                        w_eff, b_eff, out, n, L, count)
  switch (n_tasks) {
    case 1: SVDD_TAIL(1); break;
    case 2: SVDD_TAIL(2); break;
    case 3: SVDD_TAIL(3); break;
    default: SVDD_TAIL(4); break;
  }
#undef SVDD_TAIL
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_candidate_windows(const uint8_t* cand, const uint8_t* x, int B, int L, int M, int margin, int32_t* win,
                                      int32_t* flags, void* stream) {
  if (!cand || !x || !win || B <= 0 || L <= 0 || M <= 0 || margin < 0) return SVDD_E_ARG;
  const int n = B * M;
  hipLaunchKernelGGL(candidate_windows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, cand, x,
                     n, L, M, margin, win, flags);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_conv_tower_windows_f32(const float* onehot, const float* tiles, const float* bias, const int32_t* win,
                                           const float* parent_out, float* out, int n, int L, int M, int nlayers,
                                           int residual_mask, const int32_t* live_idx, const int32_t* count, void* stream) {
  if (!onehot || !tiles || !bias || !win || !parent_out || !out || n <= 0 || M <= 0 || (n % M && !live_idx) || L <= TW_ROWS / 2 ||
      L > TW_ROWS || nlayers != 5)
    return SVDD_E_ARG;                                   // one sequence per tile; margins below assume the 5-layer tower
  // (with an index list, n is only the number of list entries this launch may take: a launch over PART of a compacted list)
  TowerWinArgs wa{TowerArgs{onehot, tiles, bias, out, n, L, 1, nlayers, residual_mask, nullptr}, win, parent_out, M, live_idx, count};
  const size_t lds = sizeof(float) * ((size_t)(TW_ROWS + 2) * TW_AP + (size_t)(TW_ROWS + 16) * 4);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(5, &e0, &e1);
  if (g_tower_ver != 1) {
    auto go = [&](auto kern) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipExtLaunchKernelGGL(kern, dim3((unsigned)n), dim3(512), lds, (hipStream_t)stream, e0, e1, 0, wa);
    };
    if (g_tower_ver == 2) go(conv_tower2_kernel<true, true, 2>); else go(conv_tower2_kernel<true, true, 1>);
    return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
  }
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_tower_win_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipExtLaunchKernelGGL(conv_tower_win_kernel, dim3((unsigned)n), dim3(512), lds, (hipStream_t)stream, e0, e1, 0, wa);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}
