// kvae_lgssm_wide.hip — the SAME filter / smoother / backward bodies (lgssm_fwd.h, lgssm_bwd.h) built with FOUR
// wavefronts (256 threads) per sequence.  For n > 8 a 16x16 tile has 256 elements: one thread per output element
// instead of four per lane, and KV_SYNC() becomes a real 4-wave workgroup barrier.  At the stress shape (n = 16,
// T = 200, 512 sequences per GPU) only 512 workgroups exist, so widening each of them is what puts more of the
// chip to work; the bodies are written against KV_PAR / KV_LANES and need no change.
// Separate translation unit (own namespace) because KV_LANES is a compile-time constant of the bodies.
#define KV_LANES 256
#define kvae kvae_w256
#include <hip/hip_runtime.h>

#include "lgssm_bwd.h"
#include "lgssm_fwd.h"

using namespace kvae;

template <class D>
__global__ __launch_bounds__(256) void k_smooth_fwd_wide(kvae_lgssm_problem P, kvae_lgssm_states S, int do_filter, int do_rts) {
  __shared__ FwdLds<D> L;
  const D d(P.n, P.m, P.p);
  const int b = blockIdx.x;
  if (do_filter) {
    filter_sweep(d, P, S, b, L);
    KV_SYNC();
  }
  if (do_rts) rts_sweep(d, P, S, b, L);
}

template <class D>
__global__ __launch_bounds__(256) void k_smooth_bwd_wide(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
                                                         kvae_lgssm_input_grads G, float *ws, int with_rts) {
  __shared__ BwdLds<D> L;
  const D d(P.n, P.m, P.p);
  const int b = blockIdx.x;
  if (with_rts)
    rts_bwd_sweep(d, P, S, U, G, ws, b, L);
  else
    filter_bwd_seed(d, P, U, G, ws, b);
  KV_SYNC();
  filter_bwd_sweep(d, P, S, G, ws, b, L);
}

// launchers used by kvae_lgssm.hip (not part of the public C ABI)
extern "C" void kvae_wide_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                     hipStream_t s) {
  if (p->n == 16 && p->m == 16 && p->p == 2)
    k_smooth_fwd_wide<SDims<16, 16, 2>><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, do_filter, do_rts);
  else
    k_smooth_fwd_wide<RDims><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, do_filter, do_rts);
}
extern "C" void kvae_wide_launch_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                                     const kvae_lgssm_input_grads *out, float *ws, int with_rts, hipStream_t s) {
  if (p->n == 16 && p->m == 16 && p->p == 2)
    k_smooth_bwd_wide<SDims<16, 16, 2>><<<dim3(p->B), dim3(256), 0, s>>>(*p, *saved, *up, *out, ws, with_rts);
  else
    k_smooth_bwd_wide<RDims><<<dim3(p->B), dim3(256), 0, s>>>(*p, *saved, *up, *out, ws, with_rts);
}

// ---------------------------------------------------------------------------------------------
// Kalman filter with the alpha-network INSIDE the time loop: when frames are missing the LSTM
// input of step t+1 is C_t mu_{t|t-1} of the hidden step (kalman_filter.py:183-185), so alpha cannot be precomputed.
// One 256-thread workgroup per sequence: every thread keeps its LSTM gate row in registers (as lstm_fast.h), the head
// + softmax + mixing write the step's A|B|C straight into the filter's LDS operands, then the generic filter step
// runs on all four wavefronts.  Forward only (imputation / evaluation); training with masks keeps the per-step
// differentiable path.
// ---------------------------------------------------------------------------------------------
struct AlphaNet {
  const float *w_ih, *w_hh, *b_ih, *b_hh;   // LSTM [4H,I] [4H,H] [4H] [4H]
  const float *head_w, *head_b;             // [K,H] [K]
  const float *A, *Bm, *C;                  // mode matrices [K,n,n] [K,n,m] [K,p,n]
  int K;
};

__device__ __forceinline__ float w_sigmoid(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float w_tanh(float v) {
  const float e = __expf(-2.0f * fabsf(v));
  return copysignf((1.0f - e) / (1.0f + e), v);
}

template <class D, int H, int I>
__global__ __launch_bounds__(256) void k_filter_alpha_lstm(kvae_lgssm_problem P, kvae_lgssm_states S, AlphaNet N,
                                                           float *record, float *alpha_out) {
  constexpr int G = 4 * H, HP = (H + 3) / 4 * 4;
  static_assert(G <= 256, "one thread per LSTM gate row");
  __shared__ FwdLds<D> L;
  __shared__ __attribute__((aligned(16))) float sh_h[HP];
  __shared__ float sh_g[G], sh_x[I], sh_logit[16], sh_alpha[16];
  const D d(P.n, P.m, P.p);
  const int n = d.n(), m = d.m(), p = d.p(), nn = n * n, T = P.T, K = N.K;
  const int E = nn + n * m + p * n;
  const int b = blockIdx.x, j = threadIdx.x;
  float w[HP], wi[I], bias = 0.f, c = 0.f;
#pragma unroll
  for (int k = 0; k < HP; ++k) w[k] = (j < G && k < H) ? N.w_hh[j * H + k] : 0.f;
#pragma unroll
  for (int i = 0; i < I; ++i) wi[i] = (j < G) ? N.w_ih[j * I + i] : 0.f;
  if (j < G) bias = N.b_ih[j] + N.b_hh[j];
  if (j < HP) sh_h[j] = 0.f;
  if (j < I) sh_x[j] = 0.f;                               // y_for_dyn of step 0 is zero (kalman_filter.py:142)
  copy_in(L.mu, P.mu0 + (int64_t)b * P.mu0_sb, n);
  copy_in(L.Sig, P.Sigma0 + (int64_t)b * P.Sigma0_sb, nn);
  copy_in(L.R, P.R, p * p);
  __syncthreads();
  const bool is_g = (j >= 2 * H) && (j < 3 * H);
  for (int t = 0; t < T; ++t) {
    const int64_t q = (int64_t)b * T + t;
    // ---- LSTM cell on y_for_dyn (dyn_param.py:50-52) ----
    float acc = bias;
#pragma unroll
    for (int i = 0; i < I; ++i) acc = fmaf(wi[i], sh_x[i], acc);
#pragma unroll
    for (int k = 0; k < HP; k += 4) {
      const float4 hv = *reinterpret_cast<const float4 *>(&sh_h[k]);
      acc = fmaf(w[k], hv.x, acc); acc = fmaf(w[k + 1], hv.y, acc); acc = fmaf(w[k + 2], hv.z, acc); acc = fmaf(w[k + 3], hv.w, acc);
    }
    if (j < G) sh_g[j] = is_g ? w_tanh(acc) : w_sigmoid(acc);
    __syncthreads();
    if (j < H) {
      c = sh_g[H + j] * c + sh_g[j] * sh_g[2 * H + j];
      sh_h[j] = sh_g[3 * H + j] * w_tanh(c);
    }
    __syncthreads();
    // ---- head + softmax -> alpha_t (dyn_param.py:54-55) ----
    if (j < K) {
      float lg = N.head_b[j];
      for (int u = 0; u < H; ++u) lg = fmaf(N.head_w[j * H + u], sh_h[u], lg);
      sh_logit[j] = lg;
    }
    __syncthreads();
    if (j < K) {
      float mx = -INFINITY, sum = 0.f;
      for (int k = 0; k < K; ++k) mx = fmaxf(mx, sh_logit[k]);
      for (int k = 0; k < K; ++k) sum += expf(sh_logit[k] - mx);
      const float a = expf(sh_logit[j] - mx) / sum;
      sh_alpha[j] = a;
      alpha_out[q * K + j] = a;
    }
    __syncthreads();
    // ---- mixing straight into the filter operands (dyn_param.py:58-60) + this step's y, u, Q, mask ----
    float *rec = record + q * E;
    KV_PAR(e, E) {
      const float *base = e < nn ? N.A + e : (e < nn + n * m ? N.Bm + (e - nn) : N.C + (e - nn - n * m));
      const int stride = e < nn ? nn : (e < nn + n * m ? n * m : p * n);
      float v = 0.f;
      for (int k = 0; k < K; ++k) v = fmaf(sh_alpha[k], base[k * stride], v);
      rec[e] = v;
      if (e < nn) L.A[e] = v;
      else if (e < nn + n * m) L.Bm[e - nn] = v;
      else L.C[e - nn - n * m] = v;
    }
    copy_in(L.Q, stack_at(P.Q, b, t), nn);
    copy_in(L.y, P.Y + q * p, p);
    copy_in(L.u, P.U + q * m, m);
    const float mv = *mask_addr(P, b, t);
    KV_LANE0 { L.mk[0] = P.mask ? mv : 1.0f; }
    __syncthreads();
    filter_step_core(d, S, q, L);
    // ---- y_for_dyn for the next step: the frame if observed, else C_t mu_{t|t-1} (kalman_filter.py:183-185) ----
    if (j < I) {
      float yp = 0.f;
      for (int k = 0; k < n; ++k) yp = fmaf(L.C[j * n + k], L.mup[k], yp);
      const float mk = L.mk[0];
      sh_x[j] = mk * L.y[j] + (1.0f - mk) * yp;
    }
    __syncthreads();
  }
}

extern "C" int kvae_wide_launch_filter_alpha_lstm(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, const float *w_ih,
                                                  const float *w_hh, const float *b_ih, const float *b_hh, const float *head_w,
                                                  const float *head_b, const float *A, const float *Bm, const float *C, int K,
                                                  int H, float *record, float *alpha, hipStream_t s) {
  if (H != 50 || p->p != 2 || K < 1 || K > 16) return KVAE_ERR_DIMS;
  const AlphaNet net{w_ih, w_hh, b_ih, b_hh, head_w, head_b, A, Bm, C, K};
  if (p->n == 4 && p->m == 4)
    k_filter_alpha_lstm<SDims<4, 4, 2>, 50, 2><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, net, record, alpha);
  else
    k_filter_alpha_lstm<RDims, 50, 2><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, net, record, alpha);
  return KVAE_OK;
}
