// small_linear.h — the linear heads of the alpha-networks as HIP kernels, so that no library GEMM is left inside the captured
// LGSSM chain:  y = x W^T + b  with  F <= 128 inputs and O <= 256 outputs over N = B*T rows (reference: `head_w` + softmax,
// kvae/kalman/dyn_param.py:53-56; `linear_head` / `init_head`, kvae/kalman/switch_dyn_param.py:119-129).
//   forward            one thread per (row, group of 4 outputs) - or per row with the softmax over all O <= 16 outputs fused;
//   backward (input)   dx = gl W with gl = g, or the softmax's backward y * (g - <g, y>) (also written out: the weight
//                      gradient dW = gl^T x, db = colsum gl is rnn_wgrad.h's reduction over the same rows).
// W sits in LDS (O * F <= 12288 floats); these are HBM-trivial (a few MB) and launch-latency-sized: what matters is that they
// are ONE small launch each instead of addmm + bias + softmax (+ their backward: two GEMMs, a column sum, softmax backward).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {

constexpr int SL_MAX_W = 12288;   // floats of W in LDS (48 KiB; the staged rows of x take the rest of 60 KiB)
constexpr int SL_MAX_F = 128;
constexpr int SL_SOFTMAX_MAX_O = 16;
constexpr int SL_BATCH = 10;      // row elements fetched back to back before their FMAs (a load per FMA waits out its latency)

// W [O,F] into LDS with rows padded to F + 1 floats: threads of one row of x that work on different outputs read
// sh_w[o * (F + 1) + f] for several o at once - with an even row length (F = 50, 100) those fall into two banks
__device__ __forceinline__ void sl_stage_w_nosync(const float *__restrict__ W, float *sh, int O, int F) {
  for (int i = threadIdx.x; i < O * F; i += blockDim.x) sh[(i / F) * (F + 1) + (i % F)] = W[i];
}
__device__ __forceinline__ void sl_stage_w(const float *__restrict__ W, float *sh, int O, int F) {
  sl_stage_w_nosync(W, sh, O, F);
  __syncthreads();
}

// Rows of x reach the threads through LDS: a wavefront copies whole rows (coalesced; a thread reading ITS row straight from
// global memory touches 64 cache lines per load instruction - 18 us for 12800 x 50 floats), rows padded to F + 1 floats so that a
// thread per row reads conflict-free.
__device__ __forceinline__ void sl_stage_rows(const float *__restrict__ x, int64_t xs, int64_t n0, int64_t N, int F, int rows,
                                              float *sh_x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // rows <= 64: a wavefront owns at most 16 rows; all of their loads are issued before the first LDS write (a rolled loop keeps
  // ONE load in flight per wavefront: 26 us instead of 18 for the thread-per-row version it was meant to beat)
  for (int f0 = 0; f0 < F; f0 += 64) {
    const int f = f0 + lane;
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = wave + 4 * u;
      const int64_t n = n0 + r;
      const bool ok = r < rows && n < N && f < F;
      const float t = x[ok ? n * xs + f : 0];              // unconditional load from a valid address, select afterwards
      v[u] = ok ? t : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = wave + 4 * u;
      if (r < rows && f < F) sh_x[r * (F + 1) + f] = v[u];
    }
  }
}
__host__ __device__ inline int sl_rows_per_block(int F, int O, int64_t N) {   // what fits 60 KiB of LDS next to W, at most 64;
  const int room = (15360 - O * (F + 1)) / (F + 1);                             // fewer when that leaves CUs without a block
  int rows = room >= 64 ? 64 : (room >= 32 ? 32 : 16);
  while (rows > 16 && N / rows < 256) rows >>= 1;
  return rows;
}

// thread per (row, 4 outputs), `rows` rows per block
__global__ __launch_bounds__(256) void k_linear_fwd(const float *__restrict__ x, int64_t xs, int64_t N, int F,
                                                    const float *__restrict__ W, const float *__restrict__ b, int O,
                                                    float *__restrict__ y, int rows) {
  extern __shared__ float sh_w[];
  float *sh_x = sh_w + O * (F + 1);
  const int64_t n0 = (int64_t)blockIdx.x * rows;
  sl_stage_w_nosync(W, sh_w, O, F);
  sl_stage_rows(x, xs, n0, N, F, rows, sh_x);
  __syncthreads();
  const int OG = (O + 3) / 4;
  for (int work = threadIdx.x; work < rows * OG; work += 256) {
    const int r = work / OG, o0 = (work % OG) * 4;
    const int64_t n = n0 + r;
    if (n >= N) continue;
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (b && o0 + j < O) ? b[o0 + j] : 0.f;
    const float *xr = sh_x + r * (F + 1);
#pragma unroll 10
    for (int f = 0; f < F; ++f) {   // (unrolled: the LDS reads of ten inputs are in flight together)
      const float xv = xr[f];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (o0 + j < O) acc[j] = fmaf(xv, sh_w[(o0 + j) * (F + 1) + f], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (o0 + j < O) y[n * O + o0 + j] = acc[j];
  }
}

// thread per row, softmax over the O <= 16 outputs
__global__ __launch_bounds__(256) void k_linear_softmax_fwd(const float *__restrict__ x, int64_t xs, int64_t N, int F,
                                                            const float *__restrict__ W, const float *__restrict__ b, int O,
                                                            float *__restrict__ y, int rows) {
  extern __shared__ float sh_w[];
  float *sh_x = sh_w + O * (F + 1);
  const int64_t n0 = (int64_t)blockIdx.x * rows;
  sl_stage_w_nosync(W, sh_w, O, F);
  sl_stage_rows(x, xs, n0, N, F, rows, sh_x);
  __syncthreads();
  const int r = threadIdx.x;
  const int64_t n = n0 + r;
  if (r >= rows || n >= N) return;
  float acc[SL_SOFTMAX_MAX_O];
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o) acc[o] = (b && o < O) ? b[o] : 0.f;
  const float *xr = sh_x + r * (F + 1);
#pragma unroll 10
  for (int f = 0; f < F; ++f) {   // (unrolled: the LDS reads of ten inputs are in flight together)
    const float xv = xr[f];
#pragma unroll
    for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o)
      if (o < O) acc[o] = fmaf(xv, sh_w[o * (F + 1) + f], acc[o]);
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
  sl_stage_w(W, sh_w, O, F);
  const int FG = (F + 3) / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = idx / FG;
  const int f0 = (int)(idx % FG) * 4;
  if (n >= N) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float *gr = g + n * O;
  for (int o0 = 0; o0 < O; o0 += SL_BATCH) {
    float gv[SL_BATCH];
#pragma unroll
    for (int u = 0; u < SL_BATCH; ++u) gv[u] = o0 + u < O ? gr[o0 + u] : 0.f;
#pragma unroll
    for (int u = 0; u < SL_BATCH; ++u)
      if (o0 + u < O) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (f0 + j < F) acc[j] = fmaf(gv[u], sh_w[(o0 + u) * (F + 1) + f0 + j], acc[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (f0 + j < F) dx[n * dxs + f0 + j] = acc[j];
}

// softmax backward + dx: gl = y * (g - <g, y>), dx = gl W.  Thread per (row, 4 inputs) like the plain kernel - every thread of a
// row recomputes the row's gl (O <= 16 values: cheaper than a second launch or a trip through LDS), the first one writes it out
// for the weight-gradient reduction; a row's dx leaves as 16-byte pieces from neighbouring lanes.
__global__ __launch_bounds__(256) void k_linear_softmax_bwd_input(const float *__restrict__ g, const float *__restrict__ y,
                                                                  int64_t N, int F, const float *__restrict__ W, int O,
                                                                  float *__restrict__ gl_out, float *__restrict__ dx,
                                                                  int64_t dxs) {
  extern __shared__ float sh_w[];
  sl_stage_w(W, sh_w, O, F);
  const int FG = (F + 3) / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = idx / FG;
  const int f0 = (int)(idx % FG) * 4;
  if (n >= N) return;
  float gl[SL_SOFTMAX_MAX_O], gv[SL_SOFTMAX_MAX_O], yv[SL_SOFTMAX_MAX_O];
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o) {
    gv[o] = o < O ? g[n * O + o] : 0.f;
    yv[o] = o < O ? y[n * O + o] : 0.f;
  }
  float dot = 0.f;
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o) dot = fmaf(gv[o], yv[o], dot);
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o) {
    gl[o] = yv[o] * (gv[o] - dot);
    if (f0 == 0 && o < O) gl_out[n * O + o] = gl[o];
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int o = 0; o < SL_SOFTMAX_MAX_O; ++o)
    if (o < O) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (f0 + j < F) acc[j] = fmaf(gl[o], sh_w[o * (F + 1) + f0 + j], acc[j]);
    }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (f0 + j < F) dx[n * dxs + f0 + j] = acc[j];
}

}  // namespace kvae
