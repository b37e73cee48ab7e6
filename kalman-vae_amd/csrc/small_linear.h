// small_linear.h — the linear heads of the alpha-networks as HIP kernels, so that no library GEMM is left inside the captured
// LGSSM chain:  y = x W^T + b  with  F <= 128 inputs and O <= 256 outputs over N = B*T rows (reference: `head_w` + softmax,
// kvae/kalman/dyn_param.py:53-56; `linear_head` / `init_head`, kvae/kalman/switch_dyn_param.py:119-129).
//   forward            one thread per (row, group of 4 outputs) - or per row with the softmax over all O <= 16 outputs fused;
//   backward (input)   dx = gl W with gl = g, or the softmax's backward y * (g - <g, y>) (also written out: the weight
//                      gradient dW = gl^T x, db = colsum gl is rnn_wgrad.h's reduction over the same rows).
// W sits in LDS (O * F <= 16384 floats); these are HBM-trivial (a few MB) and launch-latency-sized: what matters is that they
// are ONE small launch each instead of addmm + bias + softmax (+ their backward: two GEMMs, a column sum, softmax backward).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {

constexpr int SL_MAX_W = 16384;   // floats of W in LDS (64 KiB)
constexpr int SL_MAX_F = 128;
constexpr int SL_SOFTMAX_MAX_O = 16;

__device__ __forceinline__ void sl_stage_w(const float *__restrict__ W, float *sh, int count) {
  for (int i = threadIdx.x; i < count; i += blockDim.x) sh[i] = W[i];
  __syncthreads();
}

// thread per (row, 4 outputs)
__global__ __launch_bounds__(256) void k_linear_fwd(const float *__restrict__ x, int64_t xs, int64_t N, int F,
                                                    const float *__restrict__ W, const float *__restrict__ b, int O,
                                                    float *__restrict__ y) {
  extern __shared__ float sh_w[];
  sl_stage_w(W, sh_w, O * F);
  const int OG = (O + 3) / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = idx / OG;
  const int o0 = (int)(idx % OG) * 4;
  if (n >= N) return;
  float acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (b && o0 + j < O) ? b[o0 + j] : 0.f;
  const float *xr = x + n * xs;
  for (int f = 0; f < F; ++f) {
    const float xv = xr[f];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (o0 + j < O) acc[j] = fmaf(xv, sh_w[(o0 + j) * F + f], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (o0 + j < O) y[n * O + o0 + j] = acc[j];
}

// thread per row, softmax over the O <= 16 outputs
__global__ __launch_bounds__(256) void k_linear_softmax_fwd(const float *__restrict__ x, int64_t xs, int64_t N, int F,
                                                            const float *__restrict__ W, const float *__restrict__ b, int O,
                                                            float *__restrict__ y) {
  extern __shared__ float sh_w[];
  sl_stage_w(W, sh_w, O * F);
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float acc[SL_SOFTMAX_MAX_O];
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o) acc[o] = (b && o < O) ? b[o] : 0.f;
  const float *xr = x + n * xs;
  for (int f = 0; f < F; ++f) {
    const float xv = xr[f];
#pragma unroll
    for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o)
      if (o < O) acc[o] = fmaf(xv, sh_w[o * F + f], acc[o]);
  }
  float mx = acc[0];
#pragma unroll
  for (int o = 1; o < SL_SOFTMAX_MAX_O; ++o)
    if (o < O) mx = fmaxf(mx, acc[o]);
  float sum = 0.f;
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o)
    if (o < O) {
      acc[o] = __expf(acc[o] - mx);
      sum += acc[o];
    }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o)
    if (o < O) y[n * O + o] = acc[o] * inv;
}

// dx[n][f0..f0+3] = sum_o g[n][o] W[o][f0..]: thread per (row, 4 inputs)
__global__ __launch_bounds__(256) void k_linear_bwd_input(const float *__restrict__ g, int64_t N, int F,
                                                          const float *__restrict__ W, int O, float *__restrict__ dx,
                                                          int64_t dxs) {
  extern __shared__ float sh_w[];
  sl_stage_w(W, sh_w, O * F);
  const int FG = (F + 3) / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = idx / FG;
  const int f0 = (int)(idx % FG) * 4;
  if (n >= N) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float *gr = g + n * O;
  for (int o = 0; o < O; ++o) {
    const float gv = gr[o];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (f0 + j < F) acc[j] = fmaf(gv, sh_w[o * F + f0 + j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (f0 + j < F) dx[n * dxs + f0 + j] = acc[j];
}

// softmax backward + dx, thread per row: gl = y * (g - <g, y>) (written out), dx = gl W
__global__ __launch_bounds__(256) void k_linear_softmax_bwd_input(const float *__restrict__ g, const float *__restrict__ y,
                                                                  int64_t N, int F, const float *__restrict__ W, int O,
                                                                  float *__restrict__ gl_out, float *__restrict__ dx,
                                                                  int64_t dxs) {
  extern __shared__ float sh_w[];
  sl_stage_w(W, sh_w, O * F);
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float gl[SL_SOFTMAX_MAX_O];
  float dot = 0.f;
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o)
    if (o < O) dot = fmaf(g[n * O + o], y[n * O + o], dot);
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o) {
    gl[o] = 0.f;
    if (o < O) {
      gl[o] = y[n * O + o] * (g[n * O + o] - dot);
      gl_out[n * O + o] = gl[o];
    }
  }
  for (int f = 0; f < F; ++f) {
    float acc = 0.f;
#pragma unroll
    for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o)
      if (o < O) acc = fmaf(gl[o], sh_w[o * F + f], acc);
    dx[n * dxs + f] = acc;
  }
}

}  // namespace kvae
