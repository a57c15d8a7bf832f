// svdd_gru_train.hip — the bidirectional GRU of the ConvGRU reward net with a BACKWARD pass to its input
// (reference Enformer.py:1595-1602 nn.GRU(64, 64, bidirectional=True, batch_first=True); used by the gradient-guidance baseline
// of BASELINE.json configs[4]: diffusion_gosai.py:1321-1330 compute_gradient_DPS differentiates reward(softmax(E[x0 | x_t]))
// with respect to the one-hot input, i.e. straight through this recurrence).
//
// Round 2/3a ran that through PyTorch autograd: per-timestep native cells (113 ms per gradient at B = 256), then MIOpen's fused
// RNN (forward + backward 35.9 ms of a 54 ms gradient, tools/dps_split.py) — against 0.4 ms for the inference kernel of
// svdd_nets.hip. The weights are frozen in every decode path, so only d/dx is needed:
//
//   svdd_gru_bidir_train_f32   the forward recurrence of gru_bidir_kernel<false> (same bits) that also SAVES, per direction,
//                              sequence, step and unit, the gates r, z, the candidate n and the recurrent pre-activation
//                              W_hn h + b_hn: what the backward pass needs besides the hidden states themselves
//   svdd_gru_bidir_bwd_f32     back-propagation through time: per (tile of 16 sequences, direction) one workgroup walks the
//                              steps in reverse; a step = gate derivatives (element-wise, in the MFMA C/D layout of the
//                              forward kernel: lane (j, g) register rho <-> sequence 4 g + rho, unit 16 w + j) -> two 16 x 192
//                              matrices in LDS -> 48 + 48 fp32 MFMAs: dh_{t-1} = [da_r | da_z | da_n r] W_hh and
//                              dx_t = [da_r | da_z | da_n] W_ih, with both weight sets resident in registers
//
// GRU equations (PyTorch):  r = s(W_ir x + b_ir + W_hr h + b_hr) ; z = s(W_iz x + b_iz + W_hz h + b_hz)
//                           n = tanh(W_in x + b_in + r (W_hn h + b_hn)) ; h' = (1 - z) n + z h
// Backward, given dh' (output gradient of the step + recurrent gradient from the step after it in walking order):
//   dn = dh' (1 - z) ; dz = dh' (h - n) ; da_n = dn (1 - n^2) ; da_r = da_n (W_hn h + b_hn) r (1 - r) ; da_z = dz z (1 - z)
//   dx = W_ir^T da_r + W_iz^T da_z + W_in^T da_n ; dh = dh' z + W_hr^T da_r + W_hz^T da_z + W_hn^T (da_n r)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "svdd_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64;          // hidden = input width
constexpr int TS = 16;         // sequences per workgroup (MFMA M)
constexpr int HPAD = H + 4;    // LDS row stride of the hidden state (floats)
constexpr int DP = 3 * H + 4;  // LDS row stride of a gate-derivative matrix (floats): 196 = 4 mod 64 banks per row

__device__ __forceinline__ float sigmoid_fast(float a) {        // as in svdd_nets.hip: the forward pass must give the same bits
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * a));
}
__device__ __forceinline__ float tanh_fast(float a) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539008177793f * a));
}

// ---- forward with saved gates. x [n, L, 64]; wpack / bpack as svdd_gru_bidir_f32 (svdd_amd.fused.pack_gru);
//      out [2][n][L][64]; save [2][n][L][4][64] = r, z, n, W_hn h + b_hn. One workgroup (4 waves) per (tile, direction).
__global__ __launch_bounds__(256) void gru_train_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wpack,
                                                            const float* __restrict__ bpack, float* __restrict__ out,
                                                            float* __restrict__ save, int n, int L) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][TS][HPAD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int dir = (int)(blockIdx.x & 1);
  const int j = lane & 15, g = lane >> 4;
  const int seq0 = (int)(blockIdx.x >> 1) * TS;
  if (seq0 >= n) return;
  float wr[96];
  {
    const float4* wp = reinterpret_cast<const float4*>(wpack + (((size_t)dir * 4 + w) * 64 + lane) * 96);
#pragma unroll
    for (int i = 0; i < 24; ++i) { const float4 v = wp[i]; wr[4 * i] = v.x; wr[4 * i + 1] = v.y; wr[4 * i + 2] = v.z; wr[4 * i + 3] = v.w; }
  }
  const int u = 16 * w + j;
  const float b_r = bpack[(dir * 4 + 0) * H + u], b_z = bpack[(dir * 4 + 1) * H + u];
  const float b_nx = bpack[(dir * 4 + 2) * H + u], b_nh = bpack[(dir * 4 + 3) * H + u];
  const int arow = min(seq0 + j, n - 1);
  const float* xrow = x + (size_t)arow * L * H + 16 * g;
  float hprev[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = threadIdx.x; i < TS * HPAD; i += 256) (&hbuf[0][0][0])[i] = 0.0f;
  const int t0 = dir == 0 ? 0 : L - 1, dt = dir == 0 ? 1 : -1;
  float xa[16], xn[16];
  {
    const float4* xp = reinterpret_cast<const float4*>(xrow + (size_t)t0 * H);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = xp[i]; xa[4 * i] = v.x; xa[4 * i + 1] = v.y; xa[4 * i + 2] = v.z; xa[4 * i + 3] = v.w; }
  }
  f32x4 acc_r = {b_r, b_r, b_r, b_r}, acc_z = {b_z, b_z, b_z, b_z}, acc_nx = {b_nx, b_nx, b_nx, b_nx};
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[s], acc_r, 0, 0, 0);
    acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[32 + s], acc_z, 0, 0, 0);
    acc_nx = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wr[64 + s], acc_nx, 0, 0, 0);
  }
  __syncthreads();
  for (int step = 0; step < L; ++step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    if (step + 1 < L) {
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
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {                // C/D layout: reg rho -> sequence 4g + rho, unit u
      const float r = sigmoid_fast(acc_r[rho]);
      const float z = sigmoid_fast(acc_z[rho]);
      const float nn = tanh_fast(acc_nx[rho] + r * acc_nh[rho]);
      const float hn = (1.0f - z) * nn + z * hprev[rho];
      hprev[rho] = hn;
      const int srow = 4 * g + rho;
      hbuf[cur ^ 1][srow][u] = hn;
      if (seq0 + srow < n) {
        const size_t at = ((size_t)dir * n + seq0 + srow) * L + t;
        out[at * H + u] = hn;
        float* sv = save + at * (4 * H) + u;
        sv[0] = r; sv[H] = z; sv[2 * H] = nn; sv[3 * H] = acc_nh[rho];
      }
    }
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

// ---- backward through time. gout [2][n][L][64] (gradient of the per-direction outputs); out / save from the forward above;
//      wpack_bwd [2 dirs][4 waves][64 lanes][96]: lane (j, g) of wave w holds, for output column c = 16 w + j and
//      k' = 48 g + s (s = 0..47; k' = gate * 64 + unit, gates r, z, n): [0:48) W_hh[k'][c], [48:96) W_ih[k'][c]
//      (the B operands of v_mfma_f32_16x16x4_f32 with the k axis permuted so that a lane's A operand is 48 contiguous floats);
//      dx [2][n][L][64]: each direction's contribution to d/dx (the caller adds the two).
__global__ __launch_bounds__(256) void gru_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ out,
                                                      const float* __restrict__ save, const float* __restrict__ wpack_bwd,
                                                      float* __restrict__ dx, int n, int L) {
  __shared__ __attribute__((aligned(16))) float dh_m[2][TS][DP];      // [da_r | da_z | da_n r]  -> dh
  __shared__ __attribute__((aligned(16))) float di_m[2][TS][DP];      // [da_r | da_z | da_n]    -> dx
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int dir = (int)(blockIdx.x & 1);
  const int j = lane & 15, g = lane >> 4;
  const int seq0 = (int)(blockIdx.x >> 1) * TS;
  if (seq0 >= n) return;
  float wh[48], wi[48];
  {
    const float4* wp = reinterpret_cast<const float4*>(wpack_bwd + (((size_t)dir * 4 + w) * 64 + lane) * 96);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const float4 a = wp[i], b = wp[12 + i];
      wh[4 * i] = a.x; wh[4 * i + 1] = a.y; wh[4 * i + 2] = a.z; wh[4 * i + 3] = a.w;
      wi[4 * i] = b.x; wi[4 * i + 1] = b.y; wi[4 * i + 2] = b.z; wi[4 * i + 3] = b.w;
    }
  }
  const int u = 16 * w + j;
  const int t0 = dir == 0 ? 0 : L - 1, dt = dir == 0 ? 1 : -1;        // the FORWARD walk; this kernel walks it backwards
  float dh_rec[4] = {0.0f, 0.0f, 0.0f, 0.0f};                         // recurrent gradient for (sequence 4 g + rho, unit u)
  for (int step = L - 1; step >= 0; --step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const int srow = 4 * g + rho;
      float da_r = 0.0f, da_z = 0.0f, da_n = 0.0f, da_hn = 0.0f, keep = 0.0f;
      if (seq0 + srow < n) {
        const size_t at = ((size_t)dir * n + seq0 + srow) * L + t;
        const float dh = gout[at * H + u] + dh_rec[rho];
        const float* sv = save + at * (4 * H) + u;
        const float r = sv[0], z = sv[H], nn = sv[2 * H], hn_lin = sv[3 * H];
        const float hp = step > 0 ? out[(at - dt) * H + u] : 0.0f;    // h of the previous forward step
        const float dn = dh * (1.0f - z);
        da_n = dn * (1.0f - nn * nn);
        da_r = da_n * hn_lin * r * (1.0f - r);
        da_z = dh * (hp - nn) * z * (1.0f - z);
        da_hn = da_n * r;
        keep = dh * z;
      }
      dh_rec[rho] = keep;
      dh_m[cur][srow][u] = da_r; dh_m[cur][srow][H + u] = da_z; dh_m[cur][srow][2 * H + u] = da_hn;
      di_m[cur][srow][u] = da_r; di_m[cur][srow][H + u] = da_z; di_m[cur][srow][2 * H + u] = da_n;
    }
    __syncthreads();                                                  // (two buffers: one barrier per step)
    f32x4 acc_h = {0.0f, 0.0f, 0.0f, 0.0f}, acc_x = {0.0f, 0.0f, 0.0f, 0.0f};
    {
      const float4* ph = reinterpret_cast<const float4*>(&dh_m[cur][j][48 * g]);
      const float4* pi = reinterpret_cast<const float4*>(&di_m[cur][j][48 * g]);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float4 a = ph[i], b = pi[i];
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wh[4 * i], acc_h, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, wi[4 * i], acc_x, 0, 0, 0);
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wh[4 * i + 1], acc_h, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, wi[4 * i + 1], acc_x, 0, 0, 0);
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wh[4 * i + 2], acc_h, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, wi[4 * i + 2], acc_x, 0, 0, 0);
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wh[4 * i + 3], acc_h, 0, 0, 0);
        acc_x = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, wi[4 * i + 3], acc_x, 0, 0, 0);
      }
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {                               // C/D layout: reg rho -> sequence 4 g + rho, column u
      const int srow = 4 * g + rho;
      dh_rec[rho] += acc_h[rho];
      if (seq0 + srow < n) dx[(((size_t)dir * n + seq0 + srow) * L + t) * H + u] = acc_x[rho];
    }
  }
}

// ---- the element-wise half of a dilated-CNN backbone layer under autograd (reference dnaconv.py:212-247 forward2, the DPS
// gradient of diffusion_gosai.py:1321-1330): feat' = relu(conv(LayerNorm(feat + tb)) + b) + feat. The convolution runs on
// svdd_conv1d_cl_f32 in both directions; everything between two convolutions is ONE pass here, forward and backward
// (PyTorch autograd ran ~9 element-wise kernels per layer: add, LayerNorm, ReLU, add, and their backward twins).
//   bb_layer_fwd_kernel   f_out = relu(y + b) + f_prev, mask = (y + b > 0), hn = LayerNorm(f_out + tb_next) gamma + beta
//                         (y == NULL: f_out = f_prev, the first layer's LayerNorm only; gamma == NULL: no LayerNorm, last layer)
//   bb_layer_bwd_kernel   G_out = G_in + LayerNorm-backward(g_hn) at h = f_in + tb, and gt_out = G_out * mask_prev: the gradient
//                         the previous layer's convolution backward takes
// One wave per row, lane owns VPL = C / 64 adjacent channels; two-pass moments like epilogue_ln_kernel / ATen.
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct BbFwdArgs { const float* y; const float* bias; const float* f_prev; const float* tb; const float* gamma; const float* beta;
                   float eps; float* f_out; uint8_t* mask; float* hn; int64_t R; int rows_per_seq; };
template <int VPL>
__global__ __launch_bounds__(256) void bb_layer_fwd_kernel(BbFwdArgs a) {
  constexpr int C = 64 * VPL;
  const int lane = threadIdx.x & 63;
  const int c0 = lane * VPL;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < a.R; r += (int64_t)gridDim.x * 4) {
    const int64_t base = r * C + c0;
    float v[VPL];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      v[i] = a.f_prev[base + i];
      if (a.y) {
        const float t = a.y[base + i] + a.bias[c0 + i];
        v[i] += fmaxf(t, 0.0f);
        a.mask[base + i] = t > 0.0f;
        a.f_out[base + i] = v[i];
      }
      if (a.gamma) { v[i] += a.tb[(r / a.rows_per_seq) * C + c0 + i]; s += v[i]; }
    }
    if (a.gamma) {
      const float mean = wave_sum64(s) * (1.0f / C);
      float q = 0.0f;
#pragma unroll
      for (int i = 0; i < VPL; ++i) { const float d = v[i] - mean; q += d * d; }
      const float rstd = rsqrtf(wave_sum64(q) * (1.0f / C) + a.eps);
#pragma unroll
      for (int i = 0; i < VPL; ++i) a.hn[base + i] = (v[i] - mean) * rstd * a.gamma[c0 + i] + a.beta[c0 + i];
    }
  }
}

struct BbBwdArgs { const float* g_hn; const float* f_in; const float* tb; const float* gamma; float eps; const float* g_in;
                   const uint8_t* mask_prev; float* g_out; float* gt_out; int64_t R; int rows_per_seq; };
template <int VPL>
__global__ __launch_bounds__(256) void bb_layer_bwd_kernel(BbBwdArgs a) {
  constexpr int C = 64 * VPL;
  const int lane = threadIdx.x & 63;
  const int c0 = lane * VPL;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < a.R; r += (int64_t)gridDim.x * 4) {
    const int64_t base = r * C + c0;
    float h[VPL], ga[VPL];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { h[i] = a.f_in[base + i] + a.tb[(r / a.rows_per_seq) * C + c0 + i]; s += h[i]; }
    const float mean = wave_sum64(s) * (1.0f / C);
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) { h[i] -= mean; q += h[i] * h[i]; }
    const float rstd = rsqrtf(wave_sum64(q) * (1.0f / C) + a.eps);
    float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      h[i] *= rstd;                                            // x-hat
      ga[i] = a.g_hn[base + i] * a.gamma[c0 + i];
      m1 += ga[i]; m2 += ga[i] * h[i];
    }
    m1 = wave_sum64(m1) * (1.0f / C);
    m2 = wave_sum64(m2) * (1.0f / C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float g = a.g_in[base + i] + rstd * (ga[i] - m1 - h[i] * m2);
      a.g_out[base + i] = g;
      if (a.gt_out) a.gt_out[base + i] = a.mask_prev[base + i] ? g : 0.0f;
    }
  }
}

// ---------------------------------------------- the rest of the reward net's gradient pass without autograd (round 6) ----
// DPS needs d mean(score) / d input of the ConvGRU reward net (reference Enformer.py:1411-1426, 2166-2173 under
// diffusion_gosai.py:1326-1329). Round 5 ran the convolutions and the GRU on hand-written kernels but left the stem, every bias /
// residual / ReLU, the tail and all their autograd twins to ~110 torch element-wise launches per step. With the kernels below (and
// svdd_conv1d_cl_f32's fused epilogue forwards, svdd_conv1d_cl_gated_f32 backwards) the whole pass is 16 launches, no autograd.


// stem: f0[p][co] = relu(b[co] + sum_t sum_c x[p + t - T/2][c] W[t][c][co]), x [n][L][4] (any real values: softmax probabilities),
// W as [T][4][64]. One wave per position (lane = co), positions in a grid-stride loop; the lane's 4 T weights live in registers.
template <int T>
__global__ __launch_bounds__(256) void reward_stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                              float* __restrict__ out, int64_t rows, int L) {
  const int lane = threadIdx.x & 63;
  float wr[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) wr[t][c] = w[(t * 4 + c) * 64 + lane];
  const float bias = b[lane];
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
    const int pos = (int)(r % L);
    float acc = bias;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int q = pos + t - T / 2;
      if (q < 0 || q >= L) continue;                           // wave-uniform
      const float4 xv = *reinterpret_cast<const float4*>(x + (r + t - T / 2) * 4);
      acc += xv.x * wr[t][0] + xv.y * wr[t][1] + xv.z * wr[t][2] + xv.w * wr[t][3];
    }
    out[r * 64 + lane] = fmaxf(acc, 0.0f);
  }
}

// transpose of the stem: dx[p][c] = sum_t sum_co W[t][c][co] g[p - (t - T/2)][co], g [n][L][64] = the gradient at the stem's
// pre-activation (already gated by its ReLU). One wave per position, lane = co, four wave reductions.
template <int T>
__global__ __launch_bounds__(256) void reward_stem_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dx,
                                                              int64_t rows, int L) {
  const int lane = threadIdx.x & 63;
  float wr[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int c = 0; c < 4; ++c) wr[t][c] = w[(t * 4 + c) * 64 + lane];
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += nw) {
    const int pos = (int)(r % L);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int q = pos - (t - T / 2);
      if (q < 0 || q >= L) continue;
      const float gv = g[(r - (t - T / 2)) * 64 + lane];
      a0 += gv * wr[t][0]; a1 += gv * wr[t][1]; a2 += gv * wr[t][2]; a3 += gv * wr[t][3];
    }
    a0 = wave_sum64(a0); a1 = wave_sum64(a1); a2 = wave_sum64(a2); a3 = wave_sum64(a3);
    if (lane == 0) *reinterpret_cast<float4*>(dx + r * 4) = make_float4(a0, a1, a2, a3);
  }
}

// g = (a + b) where f > 0, else 0: the two directions' input gradients of the GRU summed and gated by the last tower layer's ReLU
__global__ __launch_bounds__(256) void sum_gate_kernel(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ f,
                                                       float4* __restrict__ g, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 av = a[i], bv = b[i], fv = f[i];
    g[i] = make_float4(fv.x > 0.0f ? av.x + bv.x : 0.0f, fv.y > 0.0f ? av.y + bv.y : 0.0f, fv.z > 0.0f ? av.z + bv.z : 0.0f,
                       fv.w > 0.0f ? av.w + bv.w : 0.0f);
  }
}

// The tail's forward AND backward in one pass (GRUBlock's direction sum + FFN + ConvHead + mean over length and batch, Enformer.py:
// 1617, 2010-2047, 2166-2173): the loss is mean_n(mean_l(w_eff . relu(W1 LN(h_f + h_b) + b1)) + b_eff), so its gradient with respect
// to h_f + h_b needs nothing from outside the row:  dz = [z > 0] w_eff / (n L) ; dhn = W1^T dz ; ds = LN'(gamma dhn).
// One wave per row (lane = channel); W1 sits in LDS both ways ([c][k] for z, [k][c] for dhn: both conflict-free).
__global__ __launch_bounds__(256) void reward_tail_grad_kernel(const float* __restrict__ h0, const float* __restrict__ h1, const float* __restrict__ w1 /*[128][64]*/,
                                                               const float* __restrict__ b1, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const float* __restrict__ w_eff /*[128]*/, float eps, float inv_count,
                                                               float* __restrict__ g0, float* __restrict__ g1, int64_t rows) {
  constexpr int RB = 4;                // rows per wave iteration: every weight read from LDS serves RB rows
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* w1kc = sm;                    // [128][64]
  float* w1ck = sm + 128 * 64;         // [64][128]
  float* buf = w1ck + 64 * 128;        // per wave: hn [RB][64] + dz [RB][128]
  for (int e = threadIdx.x; e < 128 * 64; e += 256) {
    const float v = w1[e];
    w1kc[e] = v;
    w1ck[(e & 63) * 128 + (e >> 6)] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* hn_s = buf + wv * (RB * 192);
  float* dz_s = hn_s + RB * 64;
  const float gm = gamma[lane], bt = beta[lane];
  const float b1a = b1[lane], b1b = b1[lane + 64], wea = w_eff[lane] * inv_count, web = w_eff[lane + 64] * inv_count;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wv) * RB; r0 < rows; r0 += nw * RB) {
    float xh[RB], rstd[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int64_t r = r0 + q < rows ? r0 + q : rows - 1;
      const float s = h0[r * 64 + lane] + h1[r * 64 + lane];
      const float mean = wave_sum64(s) * (1.0f / 64);
      const float d = s - mean;
      rstd[q] = rsqrtf(wave_sum64(d * d) * (1.0f / 64) + eps);
      xh[q] = d * rstd[q];
      hn_s[q * 64 + lane] = xh[q] * gm + bt;
    }
    float za[RB], zb[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) { za[q] = b1a; zb[q] = b1b; }
#pragma unroll 4
    for (int c = 0; c < 64; ++c) {
      const float wa = w1ck[c * 128 + lane], wb = w1ck[c * 128 + 64 + lane];
#pragma unroll
      for (int q = 0; q < RB; ++q) { const float hv = hn_s[q * 64 + c]; za[q] += hv * wa; zb[q] += hv * wb; }   // hv: a broadcast read
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) { dz_s[q * 128 + lane] = za[q] > 0.0f ? wea : 0.0f; dz_s[q * 128 + lane + 64] = zb[q] > 0.0f ? web : 0.0f; }
    float dh[RB];
#pragma unroll
    for (int q = 0; q < RB; ++q) dh[q] = 0.0f;
#pragma unroll 4
    for (int k = 0; k < 128; ++k) {
      const float wk = w1kc[k * 64 + lane];
#pragma unroll
      for (int q = 0; q < RB; ++q) dh[q] += dz_s[q * 128 + k] * wk;
    }
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const float t = dh[q] * gm;
      const float m1 = wave_sum64(t) * (1.0f / 64), m2 = wave_sum64(t * xh[q]) * (1.0f / 64);
      const float ds = rstd[q] * (t - m1 - xh[q] * m2);
      if (r0 + q < rows) { g0[(r0 + q) * 64 + lane] = ds; g1[(r0 + q) * 64 + lane] = ds; }
    }
  }
}

// ------------------------------------------------ round 6: the GRU pair with the non-recurrent halves OFF the serial chain ----
// At the DPS batch (256 sequences = 16 tiles x 2 directions = 32 workgroups on 256 CUs) the two kernels above are pure latency:
// 200 dependent steps of 96 fp32 MFMAs per wave (48 for W_h h, 48 for W_i x) forwards — 386 us — and, backwards, 96 more plus 24
// global loads per lane that were requested in the step that needs them — 788 us — with 7/8 of the chip idle. Neither W_i x_t nor
// dx_t = W_i^T da_t depends on the recurrence:
//   gru_xproj_kernel     gi[dir][row][3][64] = b + W_i x for every (sequence, step) at once, on the whole chip (the SAME accumulator
//                        sequence as the in-chain version: bias, then the 16 k-steps — the chain continues it with W_h h: same bits)
//   gru_train_fwd2_kernel  the chain with 48 MFMAs per step; the next step's gi requested a step ahead
//   gru_bwd2_kernel      the chain with 48 MFMAs per step (dh only); every saved value of step - 1 requested during step; the gate
//                        derivatives da = [da_r | da_z | da_n] go to global memory ...
//   gru_dx_gate_kernel   ... and dx = da_fwd W_i,fwd + da_bwd W_i,bwd for all rows at once, gated by the ReLU of the layer below
//                        (replaces the per-direction dx tensors, their sum and the separate gate pass)
__global__ __launch_bounds__(256) void gru_xproj_kernel(const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bpack,
                                                        float* __restrict__ gi, int64_t rows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int dir = (int)blockIdx.y;
  const int j = lane & 15, g = lane >> 4;
  float wx[48];
  {
    const float4* wp = reinterpret_cast<const float4*>(wpack + (((size_t)dir * 4 + w) * 64 + lane) * 96);
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = wp[8 * gate + i]; wx[16 * gate + 4 * i] = v.x; wx[16 * gate + 4 * i + 1] = v.y; wx[16 * gate + 4 * i + 2] = v.z; wx[16 * gate + 4 * i + 3] = v.w; }
  }
  const int u = 16 * w + j;
  const float b_r = bpack[(dir * 4 + 0) * H + u], b_z = bpack[(dir * 4 + 1) * H + u], b_nx = bpack[(dir * 4 + 2) * H + u];
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * 16;
    const int64_t arow = r0 + j < rows ? r0 + j : rows - 1;
    float xa[16];
    const float4* xp = reinterpret_cast<const float4*>(x + arow * H + 16 * g);
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float4 v = xp[i]; xa[4 * i] = v.x; xa[4 * i + 1] = v.y; xa[4 * i + 2] = v.z; xa[4 * i + 3] = v.w; }
    f32x4 acc_r = {b_r, b_r, b_r, b_r}, acc_z = {b_z, b_z, b_z, b_z}, acc_nx = {b_nx, b_nx, b_nx, b_nx};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wx[s], acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wx[16 + s], acc_z, 0, 0, 0);
      acc_nx = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], wx[32 + s], acc_nx, 0, 0, 0);
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const int64_t row = r0 + 4 * g + rho;
      if (row < rows) {
        float* o = gi + ((size_t)dir * rows + row) * (3 * H) + u;
        o[0] = acc_r[rho]; o[H] = acc_z[rho]; o[2 * H] = acc_nx[rho];
      }
    }
  }
}

__global__ __launch_bounds__(256) void gru_train_fwd2_kernel(const float* __restrict__ gi, const float* __restrict__ wpack, const float* __restrict__ bpack,
                                                             float* __restrict__ out, float* __restrict__ save, int n, int L) {
  __shared__ __attribute__((aligned(16))) float hbuf[2][TS][HPAD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int dir = (int)(blockIdx.x & 1);
  const int j = lane & 15, g = lane >> 4;
  const int seq0 = (int)(blockIdx.x >> 1) * TS;
  if (seq0 >= n) return;
  float wh[48];                                                     // the recurrent halves: W_hr, W_hz, W_hn
  {
    const float4* wp = reinterpret_cast<const float4*>(wpack + (((size_t)dir * 4 + w) * 64 + lane) * 96);
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = wp[8 * gate + 4 + i]; wh[16 * gate + 4 * i] = v.x; wh[16 * gate + 4 * i + 1] = v.y; wh[16 * gate + 4 * i + 2] = v.z; wh[16 * gate + 4 * i + 3] = v.w; }
  }
  const int u = 16 * w + j;
  const float b_nh = bpack[(dir * 4 + 3) * H + u];
  const size_t rows = (size_t)n * L;
  float hprev[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = threadIdx.x; i < TS * HPAD; i += 256) (&hbuf[0][0][0])[i] = 0.0f;
  const int t0 = dir == 0 ? 0 : L - 1, dt = dir == 0 ? 1 : -1;
  const float* gbase[4];
#pragma unroll
  for (int rho = 0; rho < 4; ++rho) gbase[rho] = gi + ((size_t)dir * rows + (size_t)min(seq0 + 4 * g + rho, n - 1) * L) * (3 * H) + u;
  float cr[4], cz[4], cn[4], nr[4], nz[4], nn_[4];
#pragma unroll
  for (int rho = 0; rho < 4; ++rho) { const float* p = gbase[rho] + (size_t)t0 * (3 * H); cr[rho] = p[0]; cz[rho] = p[H]; cn[rho] = p[2 * H]; }
  __syncthreads();
  for (int step = 0; step < L; ++step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    if (step + 1 < L) {
#pragma unroll
      for (int rho = 0; rho < 4; ++rho) { const float* p = gbase[rho] + (size_t)(t + dt) * (3 * H); nr[rho] = p[0]; nz[rho] = p[H]; nn_[rho] = p[2 * H]; }
    }
    float ha[16];
    {
      const float4* hp = reinterpret_cast<const float4*>(&hbuf[cur][j][16 * g]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float4 v = hp[i]; ha[4 * i] = v.x; ha[4 * i + 1] = v.y; ha[4 * i + 2] = v.z; ha[4 * i + 3] = v.w; }
    }
    f32x4 acc_nh = {b_nh, b_nh, b_nh, b_nh}, acc_r = {cr[0], cr[1], cr[2], cr[3]}, acc_z = {cz[0], cz[1], cz[2], cz[3]};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc_nh = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wh[32 + s], acc_nh, 0, 0, 0);
      acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wh[s], acc_r, 0, 0, 0);
      acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(ha[s], wh[16 + s], acc_z, 0, 0, 0);
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const float r = sigmoid_fast(acc_r[rho]);
      const float z = sigmoid_fast(acc_z[rho]);
      const float nn = tanh_fast(cn[rho] + r * acc_nh[rho]);
      const float hn = (1.0f - z) * nn + z * hprev[rho];
      hprev[rho] = hn;
      const int srow = 4 * g + rho;
      hbuf[cur ^ 1][srow][u] = hn;
      if (seq0 + srow < n) {
        const size_t at = ((size_t)dir * n + seq0 + srow) * L + t;
        out[at * H + u] = hn;
        float* sv = save + at * (4 * H) + u;
        sv[0] = r; sv[H] = z; sv[2 * H] = nn; sv[3 * H] = acc_nh[rho];
      }
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) { cr[rho] = nr[rho]; cz[rho] = nz[rho]; cn[rho] = nn_[rho]; }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gru_bwd2_kernel(const float* __restrict__ gout, const float* __restrict__ out, const float* __restrict__ save,
                                                       const float* __restrict__ wpack_bwd, float* __restrict__ di, int n, int L) {
  __shared__ __attribute__((aligned(16))) float dh_m[2][TS][DP];      // [da_r | da_z | da_n r]  -> dh
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int dir = (int)(blockIdx.x & 1);
  const int j = lane & 15, g = lane >> 4;
  const int seq0 = (int)(blockIdx.x >> 1) * TS;
  if (seq0 >= n) return;
  float wh[48];
  {
    const float4* wp = reinterpret_cast<const float4*>(wpack_bwd + (((size_t)dir * 4 + w) * 64 + lane) * 96);
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float4 a = wp[i]; wh[4 * i] = a.x; wh[4 * i + 1] = a.y; wh[4 * i + 2] = a.z; wh[4 * i + 3] = a.w; }
  }
  const int u = 16 * w + j;
  const int t0 = dir == 0 ? 0 : L - 1, dt = dir == 0 ? 1 : -1;        // the FORWARD walk; this kernel walks it backwards
  size_t base[4];                                                      // (dir, sequence) row base; rows beyond n read the last sequence, never stored
  bool ok[4];
#pragma unroll
  for (int rho = 0; rho < 4; ++rho) { ok[rho] = seq0 + 4 * g + rho < n; base[rho] = ((size_t)dir * n + min(seq0 + 4 * g + rho, n - 1)) * L; }
  float dh_rec[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  // the saved values of the step in hand (c*) and of the one after it in walking order (p*), requested a step ahead
  float cg[4], c_r[4], c_z[4], c_n[4], c_l[4], c_h[4], pg[4], p_r[4], p_z[4], p_n[4], p_l[4], p_h[4];
#define GRU_BWD_FETCH(G, R, Z, N, LL, HP, STEP)                                                       \
  _Pragma("unroll") for (int rho = 0; rho < 4; ++rho) {                                               \
    const size_t at = base[rho] + (size_t)(t0 + dt * (STEP));                                         \
    G[rho] = gout[at * H + u];                                                                        \
    const float* sv = save + at * (4 * H) + u;                                                        \
    R[rho] = sv[0]; Z[rho] = sv[H]; N[rho] = sv[2 * H]; LL[rho] = sv[3 * H];                          \
    HP[rho] = (STEP) > 0 ? out[(at - dt) * H + u] : 0.0f;                                             \
  }
  GRU_BWD_FETCH(cg, c_r, c_z, c_n, c_l, c_h, L - 1)
  for (int step = L - 1; step >= 0; --step) {
    const int t = t0 + dt * step;
    const int cur = step & 1;
    if (step > 0) { GRU_BWD_FETCH(pg, p_r, p_z, p_n, p_l, p_h, step - 1) }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const int srow = 4 * g + rho;
      float da_r = 0.0f, da_z = 0.0f, da_n = 0.0f, da_hn = 0.0f, keep = 0.0f;
      if (ok[rho]) {
        const float dh = cg[rho] + dh_rec[rho];
        const float r = c_r[rho], z = c_z[rho], nn = c_n[rho];
        const float dn = dh * (1.0f - z);
        da_n = dn * (1.0f - nn * nn);
        da_r = da_n * c_l[rho] * r * (1.0f - r);
        da_z = dh * (c_h[rho] - nn) * z * (1.0f - z);
        da_hn = da_n * r;
        keep = dh * z;
        float* o = di + (base[rho] + t) * (3 * H) + u;
        o[0] = da_r; o[H] = da_z; o[2 * H] = da_n;
      }
      dh_rec[rho] = keep;
      dh_m[cur][srow][u] = da_r; dh_m[cur][srow][H + u] = da_z; dh_m[cur][srow][2 * H + u] = da_hn;
    }
    __syncthreads();                                                  // (two buffers: one barrier per step)
    f32x4 acc_h = {0.0f, 0.0f, 0.0f, 0.0f};
    {
      const float4* ph = reinterpret_cast<const float4*>(&dh_m[cur][j][48 * g]);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float4 a = ph[i];
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wh[4 * i], acc_h, 0, 0, 0);
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wh[4 * i + 1], acc_h, 0, 0, 0);
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wh[4 * i + 2], acc_h, 0, 0, 0);
        acc_h = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wh[4 * i + 3], acc_h, 0, 0, 0);
      }
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      dh_rec[rho] += acc_h[rho];
      cg[rho] = pg[rho]; c_r[rho] = p_r[rho]; c_z[rho] = p_z[rho]; c_n[rho] = p_n[rho]; c_l[rho] = p_l[rho]; c_h[rho] = p_h[rho];
    }
  }
#undef GRU_BWD_FETCH
}

// dx[row][c] = sum_dir sum_k' da[dir][row][k'] W_ih,dir[k'][c], gated: g = gate[row][c] > 0 ? dx : 0 (gate == NULL: no gate).
// da [2][rows][192]; the B operands are the [48:96) halves of pack_gru_bwd's images. One wave = 16 rows x 16 columns.
__global__ __launch_bounds__(256) void gru_dx_gate_kernel(const float* __restrict__ di, const float* __restrict__ wpack_bwd, const float* __restrict__ gate,
                                                          float* __restrict__ gout, int64_t rows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  float wi[2][48];
#pragma unroll
  for (int dir = 0; dir < 2; ++dir) {
    const float4* wp = reinterpret_cast<const float4*>(wpack_bwd + (((size_t)dir * 4 + w) * 64 + lane) * 96 + 48);
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float4 b = wp[i]; wi[dir][4 * i] = b.x; wi[dir][4 * i + 1] = b.y; wi[dir][4 * i + 2] = b.z; wi[dir][4 * i + 3] = b.w; }
  }
  const int u = 16 * w + j;
  const int64_t ntiles = (rows + 15) / 16;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * 16;
    const int64_t arow = r0 + j < rows ? r0 + j : rows - 1;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {
      const float4* pa = reinterpret_cast<const float4*>(di + ((size_t)dir * rows + arow) * (3 * H) + 48 * g);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float4 a = pa[i];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wi[dir][4 * i], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wi[dir][4 * i + 1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wi[dir][4 * i + 2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wi[dir][4 * i + 3], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int rho = 0; rho < 4; ++rho) {
      const int64_t row = r0 + 4 * g + rho;
      if (row < rows) {
        // (the value is moved out of the accumulator first: with `cond ? acc[rho] : 0.0f` hipcc 7.2 zeroes the AGPR that still holds
        //  acc[rho] before the compare and stores that — every gated element came out 0)
        float v = acc[rho];
        asm volatile("" : "+v"(v));
        const float gv = gate ? gate[row * H + u] : 1.0f;
        gout[row * H + u] = gv > 0.0f ? v : 0.0f;
      }
    }
  }
}

}  // namespace

extern "C" void svdd_internal_timed_events(int k, hipEvent_t* e0, hipEvent_t* e1);   // svdd_kernels.hip (profiling)

extern "C" {

int svdd_bb_layer_fwd_f32(const float* y, const float* bias, const float* f_prev, const float* tb, const float* gamma,
                          const float* beta, float eps, float* f_out, uint8_t* mask, float* hn, int64_t rows, int rows_per_seq,
                          int channels, void* stream) {
  if (!f_prev || rows <= 0 || rows_per_seq <= 0 || (y && (!bias || !f_out || !mask)) || (gamma && (!beta || !tb || !hn)) ||
      (!y && !gamma))
    return SVDD_E_ARG;
  BbFwdArgs a{y, bias, f_prev, tb, gamma, beta, eps, f_out, mask, hn, rows, rows_per_seq};
  const int64_t nb = (rows + 3) / 4;
  const dim3 grid((unsigned)(nb < 8192 ? nb : 8192));
  switch (channels) {
    case 64: hipLaunchKernelGGL(bb_layer_fwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 128: hipLaunchKernelGGL(bb_layer_fwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 256: hipLaunchKernelGGL(bb_layer_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    default: return SVDD_E_ARG;
  }
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_bb_layer_bwd_f32(const float* g_hn, const float* f_in, const float* tb, const float* gamma, float eps, const float* g_in,
                          const uint8_t* mask_prev, float* g_out, float* gt_out, int64_t rows, int rows_per_seq, int channels,
                          void* stream) {
  if (!g_hn || !f_in || !tb || !gamma || !g_in || !g_out || rows <= 0 || rows_per_seq <= 0 || ((gt_out == nullptr) != (mask_prev == nullptr)))
    return SVDD_E_ARG;
  BbBwdArgs a{g_hn, f_in, tb, gamma, eps, g_in, mask_prev, g_out, gt_out, rows, rows_per_seq};
  const int64_t nb = (rows + 3) / 4;
  const dim3 grid((unsigned)(nb < 8192 ? nb : 8192));
  switch (channels) {
    case 64: hipLaunchKernelGGL(bb_layer_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 128: hipLaunchKernelGGL(bb_layer_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    case 256: hipLaunchKernelGGL(bb_layer_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
    default: return SVDD_E_ARG;
  }
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_gru_bidir_train_f32(const float* x, const float* wpack, const float* bpack, float* out, float* save, int n, int L,
                             void* stream) {
  if (!x || !wpack || !bpack || !out || !save || n <= 0 || L <= 0) return SVDD_E_ARG;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(11, &e0, &e1);
  hipExtLaunchKernelGGL(gru_train_fwd_kernel, dim3(2 * (unsigned)((n + TS - 1) / TS)), dim3(256), 0, (hipStream_t)stream, e0, e1, 0,
                        x, wpack, bpack, out, save, n, L);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_gru_bidir_bwd_f32(const float* grad_out, const float* out, const float* save, const float* wpack_bwd, float* dx, int n,
                           int L, void* stream) {
  if (!grad_out || !out || !save || !wpack_bwd || !dx || n <= 0 || L <= 0) return SVDD_E_ARG;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(12, &e0, &e1);
  hipExtLaunchKernelGGL(gru_bwd_kernel, dim3(2 * (unsigned)((n + TS - 1) / TS)), dim3(256), 0, (hipStream_t)stream, e0, e1, 0,
                        grad_out, out, save, wpack_bwd, dx, n, L);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_reward_stem_f32(const float* x, const float* w, const float* b, float* out, int n, int L, int taps, void* stream) {
  if (!x || !w || !b || !out || n <= 0 || L <= 0 || taps != 15) return SVDD_E_ARG;
  const int64_t rows = (int64_t)n * L;
  const unsigned grid = (unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);   // latency-bound per position: many short-lived waves beat few long-lived ones (512 workgroups: 49 -> 69 us)
  hipEvent_t e0, e1;
  svdd_internal_timed_events(5, &e0, &e1);
  hipExtLaunchKernelGGL(reward_stem_fwd_kernel<15>, dim3(grid), dim3(256), 0, (hipStream_t)stream, e0, e1, 0, x, w, b, out, rows, L);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_reward_stem_bwd_f32(const float* g, const float* w, float* dx, int n, int L, int taps, void* stream) {
  if (!g || !w || !dx || n <= 0 || L <= 0 || taps != 15) return SVDD_E_ARG;
  const int64_t rows = (int64_t)n * L;
  const unsigned grid = (unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
  hipEvent_t e0, e1;
  svdd_internal_timed_events(5, &e0, &e1);
  hipExtLaunchKernelGGL(reward_stem_bwd_kernel<15>, dim3(grid), dim3(256), 0, (hipStream_t)stream, e0, e1, 0, g, w, dx, rows, L);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_sum_gate_f32(const float* a, const float* b, const float* f, float* g, int64_t count, void* stream) {
  if (!a || !b || !f || !g || count <= 0 || (count & 3)) return SVDD_E_ARG;
  const int64_t n4 = count / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(sum_gate_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(a),
                     reinterpret_cast<const float4*>(b), reinterpret_cast<const float4*>(f), reinterpret_cast<float4*>(g), n4);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_reward_tail_grad_f32(const float* h_fwd, const float* h_bwd, const float* w1, const float* b1, const float* gamma,
                              const float* beta, const float* w_eff, float eps, int n, int L, float* g_fwd, float* g_bwd, void* stream) {
  if (!h_fwd || !h_bwd || !w1 || !b1 || !gamma || !beta || !w_eff || !g_fwd || !g_bwd || n <= 0 || L <= 0) return SVDD_E_ARG;
  const int64_t rows = (int64_t)n * L;
  const size_t lds = sizeof(float) * (2 * 128 * 64 + 4 * 4 * 192);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(reward_tail_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const unsigned grid = (unsigned)((rows + 15) / 16 < 512 ? (rows + 15) / 16 : 512);   // 4 waves x 4 rows per iteration; 2 workgroups per CU (76 KB of LDS each)
  hipEvent_t e0, e1;
  svdd_internal_timed_events(7, &e0, &e1);
  hipExtLaunchKernelGGL(reward_tail_grad_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, e0, e1, 0, h_fwd, h_bwd, w1, b1, gamma, beta,
                        w_eff, eps, 1.0f / ((float)n * (float)L), g_fwd, g_bwd, rows);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_gru_bidir_train2_f32(const float* x, const float* wpack, const float* bpack, float* gi, float* out, float* save, int n, int L,
                              void* stream) {
  if (!x || !wpack || !bpack || !gi || !out || !save || n <= 0 || L <= 0) return SVDD_E_ARG;
  const int64_t rows = (int64_t)n * L;
  const int64_t tiles = (rows + 15) / 16;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(11, &e0, &e1);                          // one timed span over both launches
  hipExtLaunchKernelGGL(gru_xproj_kernel, dim3((unsigned)(tiles < 512 ? tiles : 512), 2), dim3(256), 0, (hipStream_t)stream, e0, nullptr, 0,
                        x, wpack, bpack, gi, rows);
  if (hipGetLastError() != hipSuccess) return SVDD_E_LAUNCH;
  hipExtLaunchKernelGGL(gru_train_fwd2_kernel, dim3(2 * (unsigned)((n + TS - 1) / TS)), dim3(256), 0, (hipStream_t)stream, nullptr, e1, 0,
                        (const float*)gi, wpack, bpack, out, save, n, L);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

int svdd_gru_bidir_bwd2_f32(const float* grad_out, const float* out, const float* save, const float* wpack_bwd, float* da, const float* gate,
                            float* g, int n, int L, void* stream) {
  if (!grad_out || !out || !save || !wpack_bwd || !da || !g || n <= 0 || L <= 0) return SVDD_E_ARG;
  hipEvent_t e0, e1;
  svdd_internal_timed_events(12, &e0, &e1);
  hipExtLaunchKernelGGL(gru_bwd2_kernel, dim3(2 * (unsigned)((n + TS - 1) / TS)), dim3(256), 0, (hipStream_t)stream, e0, nullptr, 0,
                        grad_out, out, save, wpack_bwd, da, n, L);
  if (hipGetLastError() != hipSuccess) return SVDD_E_LAUNCH;
  const int64_t rows = (int64_t)n * L;
  const int64_t tiles = (rows + 15) / 16;
  hipExtLaunchKernelGGL(gru_dx_gate_kernel, dim3((unsigned)(tiles < 1024 ? tiles : 1024)), dim3(256), 0, (hipStream_t)stream, nullptr, e1, 0,
                        (const float*)da, wpack_bwd, gate, g, rows);
  return hipGetLastError() == hipSuccess ? SVDD_OK : SVDD_E_LAUNCH;
}

}  // extern "C"
