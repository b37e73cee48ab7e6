// lstm_fast.h — gfx950 specialisation of the alpha-network LSTM for compile-time (H, I):
// 256 threads (4 wavefronts) per sequence, every thread keeps ITS weight row/column in registers,
// only h_t / d_pre travel through LDS (two workgroup barriers per time step).
//
//   forward : thread j < 4H owns gate row j: W_hh[j, :] (H VGPRs) + W_ih[j, :] (I VGPRs)
//   backward: thread (g = tid/64, k = tid%64) owns column k of gate block g of W_hh (k < H) or of W_ih
//             (H <= k < H+I): partial sums over one gate block, reduced across the 4 blocks through LDS
// Same arithmetic as lstm.h (the run-time-shape, host-simulated bodies); verified against
// torch.nn.LSTM in tests/test_gpu_parity.py::test_lstm_gpu.
#pragma once
#include <hip/hip_runtime.h>

namespace kvae {

__device__ __forceinline__ float fast_sigmoid(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) {
  const float e = __expf(-2.0f * fabsf(v));       // in (0,1]: no overflow
  const float t = (1.0f - e) / (1.0f + e);
  return copysignf(t, v);
}

template <int H, int I>
__global__ __launch_bounds__(256) void k_lstm_fwd_fast(const float *__restrict__ x, const float *__restrict__ w_ih,
                                                       const float *__restrict__ w_hh, const float *__restrict__ b_ih,
                                                       const float *__restrict__ b_hh, float *__restrict__ h_seq,
                                                       float *__restrict__ gates, float *__restrict__ c_seq, int T) {
  constexpr int G = 4 * H;
  constexpr int HP = (H + 3) / 4 * 4;
  static_assert(G <= 256, "one thread per gate row");
  __shared__ __attribute__((aligned(16))) float sh_h[HP];
  __shared__ float sh_g[G];
  const int b = blockIdx.x, j = threadIdx.x;
  float w[HP], wi[I], bias = 0.f, c = 0.f;
#pragma unroll
  for (int k = 0; k < HP; ++k) w[k] = (j < G && k < H) ? w_hh[j * H + k] : 0.f;
#pragma unroll
  for (int i = 0; i < I; ++i) wi[i] = (j < G) ? w_ih[j * I + i] : 0.f;
  if (j < G) bias = b_ih[j] + b_hh[j];
  if (j < HP) sh_h[j] = 0.f;
  __syncthreads();
  const bool is_g = (j >= 2 * H) && (j < 3 * H);
  for (int t = 0; t < T; ++t) {
    const int64_t q = (int64_t)b * T + t;
    float acc = bias;
#pragma unroll
    for (int i = 0; i < I; ++i) acc = fmaf(wi[i], x[q * I + i], acc);
#pragma unroll
    for (int k = 0; k < HP; k += 4) {
      const float4 hv = *reinterpret_cast<const float4 *>(&sh_h[k]);
      acc = fmaf(w[k], hv.x, acc);
      acc = fmaf(w[k + 1], hv.y, acc);
      acc = fmaf(w[k + 2], hv.z, acc);
      acc = fmaf(w[k + 3], hv.w, acc);
    }
    if (j < G) {
      const float a = is_g ? fast_tanh(acc) : fast_sigmoid(acc);
      sh_g[j] = a;
      gates[q * G + j] = a;
    }
    __syncthreads();
    if (j < H) {
      const float cn = sh_g[H + j] * c + sh_g[j] * sh_g[2 * H + j];
      const float hn = sh_g[3 * H + j] * fast_tanh(cn);
      c = cn;
      sh_h[j] = hn;
      c_seq[q * H + j] = cn;
      h_seq[q * H + j] = hn;
    }
    __syncthreads();
  }
}

template <int H, int I>
__global__ __launch_bounds__(256) void k_lstm_bwd_fast(const float *__restrict__ g_h, const float *__restrict__ gates,
                                                       const float *__restrict__ c_seq, const float *__restrict__ w_ih,
                                                       const float *__restrict__ w_hh, float *__restrict__ d_pre,
                                                       float *__restrict__ dx, int T) {
  constexpr int G = 4 * H;
  constexpr int HP = (H + 3) / 4 * 4;
  static_assert(H + I <= 64, "hidden units + inputs must fit one 64-lane column group");
  __shared__ __attribute__((aligned(16))) float sh_d[4][HP];  // d_pre of the current step, per gate block
  __shared__ float sh_part[4][64];                            // per-gate-block partial sums (dh | dx)
  const int b = blockIdx.x, g = threadIdx.x >> 6, k = threadIdx.x & 63;
  float wc[HP];
#pragma unroll
  for (int u = 0; u < HP; ++u) {
    float v = 0.f;
    if (u < H) {
      if (k < H) v = w_hh[(g * H + u) * H + k];
      else if (k < H + I) v = w_ih[(g * H + u) * I + (k - H)];
    }
    wc[u] = v;
  }
  sh_part[g][k] = 0.f;
  if (k < HP) sh_d[g][k] = 0.f;
  float dc = 0.f;
  __syncthreads();
  for (int t = T - 1; t >= 0; --t) {
    const int64_t q = (int64_t)b * T + t;
    if (g == 0) {
      const float s = sh_part[0][k] + sh_part[1][k] + sh_part[2][k] + sh_part[3][k];
      if (k < H) {
        const float ig = gates[q * G + k], fg = gates[q * G + H + k], gg = gates[q * G + 2 * H + k],
                    og = gates[q * G + 3 * H + k];
        const float ct = c_seq[q * H + k];
        const float cprev = t > 0 ? c_seq[(q - 1) * H + k] : 0.0f;
        const float tc = fast_tanh(ct);
        const float dh = g_h[q * H + k] + s;
        const float dct = dh * og * (1.0f - tc * tc) + dc;
        const float dai = dct * gg * ig * (1.0f - ig);
        const float daf = dct * cprev * fg * (1.0f - fg);
        const float dag = dct * ig * (1.0f - gg * gg);
        const float dao = dh * tc * og * (1.0f - og);
        dc = dct * fg;
        sh_d[0][k] = dai; sh_d[1][k] = daf; sh_d[2][k] = dag; sh_d[3][k] = dao;
        d_pre[q * G + k] = dai; d_pre[q * G + H + k] = daf; d_pre[q * G + 2 * H + k] = dag; d_pre[q * G + 3 * H + k] = dao;
      } else if (k < H + I && t + 1 < T) {
        dx[(q + 1) * I + (k - H)] = s;   // W_ih^T d_pre of step t+1, reduced over the 4 gate blocks
      }
    }
    __syncthreads();
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < HP; u += 4) {
      const float4 dv = *reinterpret_cast<const float4 *>(&sh_d[g][u]);
      acc = fmaf(wc[u], dv.x, acc);
      acc = fmaf(wc[u + 1], dv.y, acc);
      acc = fmaf(wc[u + 2], dv.z, acc);
      acc = fmaf(wc[u + 3], dv.w, acc);
    }
    sh_part[g][k] = acc;
    __syncthreads();
  }
  if (g == 0 && k >= H && k < H + I)
    dx[(int64_t)b * T * I + (k - H)] = sh_part[0][k] + sh_part[1][k] + sh_part[2][k] + sh_part[3][k];
}

}  // namespace kvae
