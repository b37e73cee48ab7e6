// gru_fast.h — the bidirectional GRU of the switching dynamics' regime posterior (reference:
// MarkovVariationalRegimePosterior.bigru, switch_dyn_param.py:113-129) for compile-time (H, I), same design as
// lstm_fast.h: one 256-thread workgroup per (sequence, direction), every thread's weight row (forward) or column
// (backward) in VGPRs, only h_t / gate pre-activation gradients through LDS, the whole T loop in-kernel.
// torch gate order (r, z, n):  r = s(W_ir x + b_ir + W_hr h + b_hr), z likewise,
//                              n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (1 - z) n + z h.
// MIOpen's GRU at B=256, T=50, H=50 costs ~7 ms per training step (forward + backward); this is two launches.
#pragma once
#include <hip/hip_runtime.h>

#include "lstm_fast.h"

namespace kvae {

struct GruWeights { const float *w_ih, *w_hh, *b_ih, *b_hh; };   // [3H,I] [3H,H] [3H] [3H]

// grid (B, 2): blockIdx.y = 0 forward in time, 1 reverse.  h_seq [B,T,2H] (direction d fills columns d*H..),
// gates [2,B,T,4H] = (r, z, n, hn) with hn = W_hn h + b_hn (needed by the backward).
template <int H, int I>
__global__ __launch_bounds__(256) void k_gru_fwd_fast(const float *__restrict__ x, GruWeights wf, GruWeights wb,
                                                      float *__restrict__ h_seq, float *__restrict__ gates, int B, int T) {
  constexpr int G = 3 * H;
  constexpr int HP = (H + 3) / 4 * 4;
  static_assert(G <= 256, "one thread per gate row");
  __shared__ __attribute__((aligned(16))) float sh_h[HP];
  __shared__ float sh_r[H], sh_z[H], sh_xn[H], sh_hn[H];
  const int b = blockIdx.x, dir = blockIdx.y, j = threadIdx.x;
  const GruWeights w = dir ? wb : wf;
  float wr[HP], wi[I], bi = 0.f, bh = 0.f, hprev = 0.f;
#pragma unroll
  for (int k = 0; k < HP; ++k) wr[k] = (j < G && k < H) ? w.w_hh[j * H + k] : 0.f;
#pragma unroll
  for (int i = 0; i < I; ++i) wi[i] = (j < G) ? w.w_ih[j * I + i] : 0.f;
  if (j < G) { bi = w.b_ih[j]; bh = w.b_hh[j]; }
  if (j < HP) sh_h[j] = 0.f;
  __syncthreads();
  float *gt = gates + (int64_t)dir * B * T * 4 * H;
  for (int s = 0; s < T; ++s) {
    const int t = dir ? T - 1 - s : s;
    const int64_t q = (int64_t)b * T + t;
    float xi = bi, hh = bh;
#pragma unroll
    for (int i = 0; i < I; ++i) xi = fmaf(wi[i], x[q * I + i], xi);
#pragma unroll
    for (int k = 0; k < HP; k += 4) {
      const float4 hv = *reinterpret_cast<const float4 *>(&sh_h[k]);
      hh = fmaf(wr[k], hv.x, hh);
      hh = fmaf(wr[k + 1], hv.y, hh);
      hh = fmaf(wr[k + 2], hv.z, hh);
      hh = fmaf(wr[k + 3], hv.w, hh);
    }
    if (j < H) {
      const float r = fast_sigmoid(xi + hh);
      sh_r[j] = r;
      gt[q * 4 * H + j] = r;
    } else if (j < 2 * H) {
      const float z = fast_sigmoid(xi + hh);
      sh_z[j - H] = z;
      gt[q * 4 * H + j] = z;
    } else if (j < G) {
      sh_xn[j - 2 * H] = xi;
      sh_hn[j - 2 * H] = hh;
      gt[q * 4 * H + 3 * H + (j - 2 * H)] = hh;
    }
    __syncthreads();
    if (j < H) {
      const float n = fast_tanh(sh_xn[j] + sh_r[j] * sh_hn[j]);
      const float z = sh_z[j];
      const float hn = (1.0f - z) * n + z * hprev;
      hprev = hn;
      sh_h[j] = hn;
      gt[q * 4 * H + 2 * H + j] = n;
      h_seq[q * 2 * H + dir * H + j] = hn;
    }
    __syncthreads();
  }
}

// BPTT. g_h [B,T,2H]; outputs d_pre_i, d_pre_h [2,B,T,3H] (gradients w.r.t. W_ih x + b_ih and W_hh h + b_hh),
// dx [2,B,T,I] (per direction; the caller adds the two).
template <int H, int I>
__global__ __launch_bounds__(192) void k_gru_bwd_fast(const float *__restrict__ g_h, const float *__restrict__ gates,
                                                      const float *__restrict__ h_seq, GruWeights wf, GruWeights wb,
                                                      float *__restrict__ d_pre_i, float *__restrict__ d_pre_h,
                                                      float *__restrict__ dx, int B, int T) {
  constexpr int G = 3 * H;
  constexpr int HP = (H + 3) / 4 * 4;
  static_assert(H + I <= 64, "hidden units + inputs must fit one 64-lane column group");
  __shared__ __attribute__((aligned(16))) float sh_dh[3][HP];   // d_pre_h of the current step, per gate block
  __shared__ __attribute__((aligned(16))) float sh_di[3][HP];   // d_pre_i of the current step
  __shared__ float sh_part[3][64];
  const int b = blockIdx.x, dir = blockIdx.y, g = threadIdx.x >> 6, k = threadIdx.x & 63;
  const GruWeights w = dir ? wb : wf;
  float wc[HP];
#pragma unroll
  for (int u = 0; u < HP; ++u) {
    float v = 0.f;
    if (u < H) {
      if (k < H) v = w.w_hh[(g * H + u) * H + k];
      else if (k < H + I) v = w.w_ih[(g * H + u) * I + (k - H)];
    }
    wc[u] = v;
  }
  sh_part[g][k] = 0.f;
  if (k < HP) { sh_dh[g][k] = 0.f; sh_di[g][k] = 0.f; }
  float dh_direct = 0.f;   // z * dh carried by the thread that owns unit k (g == 0)
  __syncthreads();
  const int64_t off = (int64_t)dir * B * T;
  for (int s = T - 1; s >= 0; --s) {
    const int t = dir ? T - 1 - s : s;
    const int tprev = dir ? t + 1 : t - 1;             // where h_{prev} of this step lives (s - 1)
    const int tnext = dir ? t - 1 : t + 1;             // the step processed just before in this loop (s + 1)
    const int64_t q = (int64_t)b * T + t;
    if (g == 0) {
      const float sum = sh_part[0][k] + sh_part[1][k] + sh_part[2][k];
      if (k < H) {
        const float *gt = gates + (off + q) * 4 * H;
        const float r = gt[k], z = gt[H + k], n = gt[2 * H + k], hn = gt[3 * H + k];
        const float hp = s > 0 ? h_seq[((int64_t)b * T + tprev) * 2 * H + dir * H + k] : 0.0f;
        const float dh = g_h[q * 2 * H + dir * H + k] + sum + dh_direct;
        const float dn = dh * (1.0f - z);
        const float dz = dh * (hp - n);
        dh_direct = dh * z;
        const float dan = dn * (1.0f - n * n);
        const float dar = dan * hn * r * (1.0f - r);
        const float daz = dz * z * (1.0f - z);
        sh_di[0][k] = dar; sh_di[1][k] = daz; sh_di[2][k] = dan;
        sh_dh[0][k] = dar; sh_dh[1][k] = daz; sh_dh[2][k] = dan * r;
        float *pi = d_pre_i + (off + q) * G, *ph = d_pre_h + (off + q) * G;
        pi[k] = dar; pi[H + k] = daz; pi[2 * H + k] = dan;
        ph[k] = dar; ph[H + k] = daz; ph[2 * H + k] = dan * r;
      } else if (k < H + I && s + 1 < T) {
        dx[(off + (int64_t)b * T + tnext) * I + (k - H)] = sum;   // W_ih^T d_pre_i of the step handled before
      }
    }
    __syncthreads();
    float acc = 0.f;
    const float *src = (k < H) ? sh_dh[g] : sh_di[g];
#pragma unroll
    for (int u = 0; u < HP; u += 4) {
      const float4 dv = *reinterpret_cast<const float4 *>(&src[u]);
      acc = fmaf(wc[u], dv.x, acc);
      acc = fmaf(wc[u + 1], dv.y, acc);
      acc = fmaf(wc[u + 2], dv.z, acc);
      acc = fmaf(wc[u + 3], dv.w, acc);
    }
    sh_part[g][k] = acc;
    __syncthreads();
  }
  if (g == 0 && k >= H && k < H + I) {
    const int t0 = dir ? T - 1 : 0;
    dx[(off + (int64_t)b * T + t0) * I + (k - H)] = sh_part[0][k] + sh_part[1][k] + sh_part[2][k];
  }
}

}  // namespace kvae
