// rnn_wgrad.h — parameter gradients of the recurrent alpha-networks and their linear heads as ONE reduction kernel family:
//
//     G[r][c] = sum_q D[q][r] * X[q][c],      q over the N = B*T (sequence, step) rows,
//     X[q]    = [ h[q + shift] (H columns, zero outside the sequence) | x[q] (I columns) | 1 (bias column) ]
//
// i.e. dW_hh | dW_ih | db of an LSTM / GRU direction from the d_pre rows its BPTT kernel wrote (reference: the autograd of
// nn.LSTM in kvae/kalman/dyn_param.py:50-56 and of nn.GRU + the two heads in kvae/kalman/switch_dyn_param.py:113-129), and
// dW | db of a linear head.  Round 2 left these to rocBLAS: 2 x 150 us per step for a 200 x 50 output with a 12800-deep
// reduction (one workgroup walks the whole K dimension) - the largest GPU-time row of the configs[1] step.
//
// Mapping (gfx950, v_mfma_f32_16x16x4_f32 - exact fp32, a k-ordered fmaf chain): the reduction index q is the MFMA K dimension,
// 16 gradient rows x 16 gradient columns per accumulator tile.  A workgroup (4 wavefronts) owns one chunk of q (split-K over
// the grid) and up to four 16-column tiles, one per wavefront; a wavefront keeps ALL row tiles of its column tile in accumulators
// (<= 16 tiles = 64 VGPRs), so the D rows are read once per wavefront (coalesced 64-byte segments, L1-shared by the four
// wavefronts) and X once per workgroup.  Operands are prefetched one k-step ahead.  Stage two sums the chunk partials in a
// FIXED order (run-to-run identical) and scatters the columns to dW_h | dW_x | db.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {

constexpr int WG_MAX_ROW_TILES = 16;    // R <= 256 gradient rows
constexpr int WG_MAX_CHUNKS = 128;      // split-K partials

struct WgradProblem {
  const float *d;      // [N, R] rows of d_pre / g_logit, row stride d_stride
  const float *h;      // hidden sequence (may be NULL: H = 0), row stride h_stride; X[q][c<H] = h[q + shift][c]
  const float *x;      // inputs (may be NULL: I = 0), row stride x_stride
  float *g_wh, *g_wx, *g_b;   // outputs [R,H], [R,I], [R] (each may be NULL)
  int64_t d_stride, h_stride, x_stride;
  int64_t N;           // rows
  int32_t R, H, I, bias;      // bias: 1 = append the column of ones
  int32_t T, shift;    // sequence length and time shift of the hidden operand (-1: h_{t-1}, +1: h_{t+1}, 0: h_t)
};
struct WgradBatch {
  WgradProblem p[4];
  int32_t n;
};

__device__ __forceinline__ float wg_x_operand(const WgradProblem &P, int64_t q, int c) {
  if (q >= P.N) return 0.f;
  if (c < P.H) {
    const int t = (int)(q % P.T) + P.shift;
    return (t >= 0 && t < P.T) ? P.h[(q + P.shift) * P.h_stride + c] : 0.f;
  }
  if (c < P.H + P.I) return P.x[q * P.x_stride + (c - P.H)];
  return (P.bias && c == P.H + P.I) ? 1.0f : 0.f;
}

// grid = (chunks, column groups of 4 tiles, problems); block = 256.  partials[problem][chunk][R][Cpad] with Cpad = 16 * col tiles
template <int RT>
__global__ __launch_bounds__(256) void k_rnn_wgrad_partial(const WgradBatch batch, float *__restrict__ partials,
                                                           int64_t partial_stride, int chunks) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const WgradProblem &P = batch.p[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int C = P.H + P.I + (P.bias ? 1 : 0), col_tiles = (C + 15) / 16, row_tiles = (P.R + 15) / 16;
  const int ct = blockIdx.y * 4 + wave;
  if (ct >= col_tiles) return;                               // whole wavefront: no barrier in this kernel
  const int64_t per = ((P.N + chunks - 1) / chunks + 3) / 4 * 4;
  const int64_t q_lo = (int64_t)blockIdx.x * per, q_hi = q_lo + per < P.N ? q_lo + per : P.N;
  const int kk = lane >> 4, nn = lane & 15, c = ct * 16 + nn;
  f4 acc[RT];
#pragma unroll
  for (int r = 0; r < RT; ++r) acc[r] = (f4){0.f, 0.f, 0.f, 0.f};
  float a_cur[RT], a_nxt[RT], b_cur, b_nxt;
  auto load = [&](int64_t q0, float (&a)[RT], float &b) {
    const int64_t q = q0 + kk;
    const bool ok = q < q_hi;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      const int row = r * 16 + nn;
      a[r] = (ok && r < row_tiles && row < P.R) ? P.d[q * P.d_stride + row] : 0.f;
    }
    b = ok ? wg_x_operand(P, q, c) : 0.f;
  };
  if (q_lo < q_hi) load(q_lo, a_cur, b_cur);
  for (int64_t q0 = q_lo; q0 < q_hi; q0 += 4) {
    if (q0 + 4 < q_hi) load(q0 + 4, a_nxt, b_nxt);
#pragma unroll
    for (int r = 0; r < RT; ++r)
      if (r < row_tiles) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[r], b_cur, acc[r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < RT; ++r) a_cur[r] = a_nxt[r];
    b_cur = b_nxt;
  }
  // accumulator layout: lane (n = lane & 15, g = lane >> 4), register i holds tile[4 g + i][n]
  float *out = partials + (int64_t)blockIdx.z * partial_stride + (int64_t)blockIdx.x * P.R * (col_tiles * 16);
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    if (r >= row_tiles) continue;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r * 16 + 4 * kk + i;
      if (row < P.R) out[(int64_t)row * (col_tiles * 16) + c] = acc[r][i];
    }
  }
}

// one thread per gradient element: chunk partials summed in chunk order
__global__ __launch_bounds__(256) void k_rnn_wgrad_final(const WgradBatch batch, const float *__restrict__ partials,
                                                         int64_t partial_stride, int chunks) {
  const WgradProblem &P = batch.p[blockIdx.y];
  const int C = P.H + P.I + (P.bias ? 1 : 0), Cpad = (C + 15) / 16 * 16;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= P.R * C) return;
  const int row = e / C, c = e % C;
  const float *src = partials + (int64_t)blockIdx.y * partial_stride + (int64_t)row * Cpad + c;
  float s = 0.f;
  for (int k = 0; k < chunks; ++k) s += src[(int64_t)k * P.R * Cpad];
  if (c < P.H) {
    if (P.g_wh) P.g_wh[(int64_t)row * P.H + c] = s;
  } else if (c < P.H + P.I) {
    if (P.g_wx) P.g_wx[(int64_t)row * P.I + (c - P.H)] = s;
  } else if (P.g_b) {
    P.g_b[row] = s;
  }
}

}  // namespace kvae
