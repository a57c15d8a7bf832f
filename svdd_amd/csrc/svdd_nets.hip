// svdd_nets.hip — gfx950 kernels for the value-network internals that PyTorch-ROCm/MIOpen runs
// badly at SVDD's shapes (B*M = 2560 short sequences, hidden 64). Exposed through the C ABI of
// include/svdd_hip.h ("net kernels" section); used by svdd_amd/fused.py.
//
//   gru_bidir_kernel   bidirectional GRU layer (input 64 -> hidden 64), fp32, one launch for all
//                      L timesteps. MIOpen's RNN path issues ~10 tiny kernels per timestep per
//                      direction (~2900 launches and 11.7 ms per value forward at n=2560, L=200
//                      [rocprof r01_v0]); here each workgroup owns 16 sequences of one direction,
//                      keeps the 96 KB of gate weights in registers (96 VGPRs/lane as MFMA B
//                      operands), the hidden state in LDS, and runs the recurrence on the exact-fp32
//                      matrix cores (v_mfma_f32_16x16x4_f32, 96 per wave-step).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "svdd_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
__global__ __launch_bounds__(256) void gru_bidir_kernel(const float* __restrict__ x, const float* __restrict__ wpack,
                                                        const float* __restrict__ bpack, float* __restrict__ out,
                                                        int n, int L) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][TS][HPAD];
  const int lane = threadIdx.x & 63;
  const int w = threadIdx.x >> 6;
  const int dir = blockIdx.y;
  const int j = lane & 15;            // hidden unit within the wave's 16 / sequence row for A operands
  const int g = lane >> 4;            // k-group of A/B operands ; row-group of C/D
  const int seq0 = blockIdx.x * TS;

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
  const int arow = min(seq0 + j, n - 1);
  const float* xrow = x + (size_t)arow * L * H + 16 * g;
  float hprev[4] = {0.0f, 0.0f, 0.0f, 0.0f};   // h[seq = 4g + rho][u] of the previous step

  // h_0 = 0
  for (int i = threadIdx.x; i < TS * HPAD; i += 256) (&hbuf[0][0][0])[i] = 0.0f;

  const int t0 = dir == 0 ? 0 : L - 1;
  const int dt = dir == 0 ? 1 : -1;
  float xa[16], xn[16];
  {
    const float4* xp = reinterpret_cast<const float4*>(xrow + (size_t)t0 * H);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = xp[i]; xa[4 * i] = v.x; xa[4 * i + 1] = v.y; xa[4 * i + 2] = v.z; xa[4 * i + 3] = v.w; }
  }
  __syncthreads();

  for (int step = 0; step < L; ++step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    if (step + 1 < L) {             // prefetch x_{t+1} under the MFMAs
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
    f32x4 acc_r = {b_r, b_r, b_r, b_r}, acc_z = {b_z, b_z, b_z, b_z};
    f32x4 acc_nx = {b_nx, b_nx, b_nx, b_nx}, acc_nh = {b_nh, b_nh, b_nh, b_nh};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[s], acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[32 + s], acc_z, 0, 0, 0);
      acc_nx = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[64 + s], acc_nx, 0, 0, 0);
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
      if (seq0 + srow < n) out[(((size_t)dir * n + seq0 + srow) * L + t) * H + u] = hn;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) xa[i] = xn[i];
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

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

}  // namespace

extern "C" int svdd_gru_bidir_f32(const float* x, const float* wpack, const float* bpack, float* out, int n, int L,
                                  void* stream) {
  if (!x || !wpack || !bpack || !out || n <= 0 || L <= 0) return SVDD_E_ARG;
  hipLaunchKernelGGL(gru_bidir_kernel, dim3((unsigned)((n + TS - 1) / TS), 2), dim3(256), 0, (hipStream_t)stream,
                     x, wpack, bpack, out, n, L);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

extern "C" int svdd_epilogue_ln_f32(const float* y, const float* bias, const float* f_prev, const float* tb,
                                    const float* gamma, const float* beta, float* f_out, float* hn, int64_t rows,
                                    int channels, int act, void* stream) {
  if (!y || (!f_out && !hn) || rows <= 0 || (hn && (!gamma || !beta)) || act < 0 || act > 2) return SVDD_E_ARG;
  EpiArgs a{y, bias, f_prev, tb, gamma, beta, f_out, hn, rows, act};
  const int64_t nblocks = (rows + 3) / 4;
  const unsigned grid = (unsigned)(nblocks < 4096 ? nblocks : 4096);
  if (channels == 64) hipLaunchKernelGGL(epilogue_ln_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (channels == 128) hipLaunchKernelGGL(epilogue_ln_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else if (channels == 256) hipLaunchKernelGGL(epilogue_ln_kernel<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  else return SVDD_E_ARG;
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}
