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

// launchers used by kvae_lgssm.hip (not part of the public C ABI)
extern "C" void kvae_wide_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                     hipStream_t s) {
  if (p->n == 16 && p->m == 16 && p->p == 2)
    k_smooth_fwd_wide<SDims<16, 16, 2>><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, do_filter, do_rts);
  else
    k_smooth_fwd_wide<RDims><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, do_filter, do_rts);
}

// ---------------------------------------------------------------------------------------------
// Kalman filter with the alpha-network INSIDE the time loop: when frames are missing the LSTM
// input of step t+1 is C_t mu_{t|t-1} of the hidden step (kalman_filter.py:183-185), so alpha cannot be precomputed.
// One 256-thread workgroup per sequence: every thread keeps its LSTM gate row in registers (as lstm_fast.h), the head
// + softmax + mixing write the step's A|B|C straight into the filter's LDS operands, then the generic filter step
// runs on all four wavefronts.  With `save` pointers the cell's internals are kept for k_alpha_lstm_bwd below, which
// makes masked TRAINING two launches (forward, backward) instead of T cell steps + T single-step filter launches.
// ---------------------------------------------------------------------------------------------
struct AlphaNet {
  const float *w_ih, *w_hh, *b_ih, *b_hh;   // LSTM [4H,I] [4H,H] [4H] [4H]
  const float *head_w, *head_b;             // [K,H] [K]
  const float *A, *Bm, *C;                  // mode matrices [K,n,n] [K,n,m] [K,p,n]
  int K;
};
struct AlphaSave {   // all optional (NULL): gates [B,T,4H] post-activation, c_seq / h_seq [B,T,H], x_seq [B,T,I] (cell inputs)
  float *gates, *c_seq, *h_seq, *x_seq;
};

__device__ __forceinline__ float w_sigmoid(float v) { return 1.0f / (1.0f + __expf(-v)); }
__device__ __forceinline__ float w_tanh(float v) {
  const float e = __expf(-2.0f * fabsf(v));
  return copysignf((1.0f - e) / (1.0f + e), v);
}

template <class D, int H, int I>
__global__ __launch_bounds__(256) void k_filter_alpha_lstm(kvae_lgssm_problem P, kvae_lgssm_states S, AlphaNet N,
                                                           float *record, float *alpha_out, AlphaSave sv) {
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
    if (j < G) {
      const float a = is_g ? w_tanh(acc) : w_sigmoid(acc);
      sh_g[j] = a;
      if (sv.gates) sv.gates[q * G + j] = a;
    }
    if (j < I && sv.x_seq) sv.x_seq[q * I + j] = sh_x[j];
    __syncthreads();
    if (j < H) {
      c = sh_g[H + j] * c + sh_g[j] * sh_g[2 * H + j];
      const float hn = sh_g[3 * H + j] * w_tanh(c);
      sh_h[j] = hn;
      if (sv.c_seq) sv.c_seq[q * H + j] = c;
      if (sv.h_seq) sv.h_seq[q * H + j] = hn;
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
                                                  int H, float *record, float *alpha, float *gates, float *c_seq, float *h_seq,
                                                  float *x_seq, hipStream_t s) {
  if (H != 50 || p->p != 2 || K < 1 || K > 16) return KVAE_ERR_DIMS;
  const AlphaNet net{w_ih, w_hh, b_ih, b_hh, head_w, head_b, A, Bm, C, K};
  const AlphaSave sv{gates, c_seq, h_seq, x_seq};
  if (p->n == 4 && p->m == 4)
    k_filter_alpha_lstm<SDims<4, 4, 2>, 50, 2><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, net, record, alpha, sv);
  else
    k_filter_alpha_lstm<RDims, 50, 2><<<dim3(p->B), dim3(256), 0, s>>>(*p, *st, net, record, alpha, sv);
  return KVAE_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the filter-with-alpha-network (and of the RTS smoother on top of it, with_rts): ONE launch.
// The two recursions are coupled in both directions - alpha_t -> (A_t, B_t, C_t) -> belief, and on hidden steps
// C_t mu_{t|t-1} -> x_{t+1} -> cell (kalman_filter.py:183-185) - so their adjoints are interleaved step by step,
// t = T-1..0:
//   1. feedback: g(C_t mu_p,t) = (1 - mask_t) gx_{t+1}  ->  the hand-off adjoint of mu_p,t gets C_t^T of it
//   2. one step of the filter adjoint (lgssm_bwd.h)      ->  gA_t, gB_t, gC_t, gY_t, gU_t, carried (gmu, gSig)
//   3. gC_t += g(C mu_p) mu_p^T ; gY_t += mask_t gx_{t+1} ; g_record_t = (gA|gB|gC)_t + upstream
//   4. mixing: g alpha_t[k] = <g_record_t, base_k> ; softmax + head: g logit_t, g h_t
//   5. LSTM cell adjoint (as lstm_fast.h)                ->  d_pre_t, carried (dh, dc), gx_t
// Parameter gradients are reductions over (b,t) of what this launch writes: g_record (-> mode matrices, kvae_mix_bwd),
// d_pre (-> LSTM weights), g_logit (-> head).
// ---------------------------------------------------------------------------------------------
struct AlphaBwd {
  const float *alpha, *gates, *c_seq;        // saved by the forward
  const float *g_record_up, *g_alpha_up;     // upstream gradients of the record / alpha outputs (NULL = none)
  float *g_record;                           // [B,T,E]  (the gA|gB|gC stacks of kvae_lgssm_input_grads point into it)
  float *d_pre, *g_logit;                    // [B,T,4H], [B,T,K]
};

template <class D, int H, int I>
__global__ __launch_bounds__(256) void k_alpha_lstm_bwd(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
                                                        kvae_lgssm_input_grads G, float *ws, int with_rts, AlphaNet N,
                                                        AlphaBwd W) {
  constexpr int GH = 4 * H, HP = (H + 3) / 4 * 4;
  static_assert(H + I <= 64, "hidden units + inputs must fit one 64-lane column group");
  __shared__ BwdLds<D> L;
  __shared__ __attribute__((aligned(16))) float sh_d[4][HP];   // d_pre of the current step, per gate block
  __shared__ float sh_part[4][64];                              // per-gate-block partial sums (dh | dx)
  __shared__ float sh_gyp[I], sh_ga[16], sh_gl[16], sh_red[4];
  const D d(P.n, P.m, P.p);
  const int n = d.n(), m = d.m(), p = d.p(), nn = n * n, T = P.T, K = N.K, rec = 2 * (n + nn);
  const int E = nn + n * m + p * n;
  const int b = blockIdx.x, tid = threadIdx.x, g = tid >> 6, k = tid & 63;
  const int64_t bT = (int64_t)b * T;
  float wc[HP];   // column k of gate block g of W_hh (k < H) or of W_ih (H <= k < H+I)
#pragma unroll
  for (int u = 0; u < HP; ++u) {
    float v = 0.f;
    if (u < H) {
      if (k < H) v = N.w_hh[(g * H + u) * H + k];
      else if (k < H + I) v = N.w_ih[(g * H + u) * I + (k - H)];
    }
    wc[u] = v;
  }
  sh_part[g][k] = 0.f;
  if (k < HP) sh_d[g][k] = 0.f;
  float dc = 0.f;
  if (with_rts)
    rts_bwd_sweep(d, P, S, U, G, ws, b, L);
  else
    filter_bwd_seed(d, P, U, G, ws, b);
  KV_SYNC();
  filter_bwd_begin(d, P, L);
  for (int t = T - 1; t >= 0; --t) {
    const int64_t q = bT + t;
    // ---- 1. feedback of x_{t+1} into the predicted mean of step t ----
    const float mv = *mask_addr(P, b, t);
    const float mk = P.mask ? mv : 1.0f;
    if (tid < I) sh_gyp[tid] = (1.0f - mk) * (sh_part[0][H + tid] + sh_part[1][H + tid] + sh_part[2][H + tid] + sh_part[3][H + tid]);
    __syncthreads();
    if (tid < n) {
      const float *Ct = stack_at(P.C, b, t);
      float acc = 0.f;
      for (int c = 0; c < I; ++c) acc = fmaf(Ct[c * n + tid], sh_gyp[c], acc);
      ws[q * rec + n + nn + tid] += acc;
    }
    __syncthreads();
    // ---- 2. filter adjoint of step t ----
    filter_bwd_step(d, P, S, G, ws, b, t, L);
    // ---- 3. complete gC_t, gY_t; total gradient of the step record ----
    if (tid < I * n) {
      const int c = tid / n, jj = tid - c * n;
      gstack_at(G.gC, b, t)[tid] += sh_gyp[c] * L.mup[jj];
    }
    if (tid < I) G.gY[q * p + tid] += mk * (sh_part[0][H + tid] + sh_part[1][H + tid] + sh_part[2][H + tid] + sh_part[3][H + tid]);
    __syncthreads();
    float *gr = W.g_record + q * E;
    if (W.g_record_up) {
      const float *up = W.g_record_up + q * E;
      for (int e = tid; e < E; e += 256) gr[e] += up[e];
      __syncthreads();
    }
    // ---- 4. mixing adjoint g alpha[kk] = <g_record, base_kk>, then softmax + head ----
    for (int kk = 0; kk < K; ++kk) {
      float part = 0.f;
      for (int e = tid; e < E; e += 256) {
        const float *base = e < nn ? N.A + (int64_t)kk * nn + e
                                   : (e < nn + n * m ? N.Bm + (int64_t)kk * n * m + (e - nn) : N.C + (int64_t)kk * p * n + (e - nn - n * m));
        part = fmaf(gr[e], *base, part);
      }
      for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
      if (k == 0) sh_red[g] = part;
      __syncthreads();
      if (tid == 0) sh_ga[kk] = (sh_red[0] + sh_red[1]) + (sh_red[2] + sh_red[3]) + (W.g_alpha_up ? W.g_alpha_up[q * K + kk] : 0.f);
      __syncthreads();
    }
    if (tid < K) {
      float dot = 0.f;
      for (int kk = 0; kk < K; ++kk) dot = fmaf(W.alpha[q * K + kk], sh_ga[kk], dot);
      const float gl = W.alpha[q * K + tid] * (sh_ga[tid] - dot);
      sh_gl[tid] = gl;
      W.g_logit[q * K + tid] = gl;
    }
    __syncthreads();
    // ---- 5. LSTM cell adjoint (cf. k_lstm_bwd_fast) ----
    if (g == 0 && k < H) {
      float gh = 0.f;
      for (int kk = 0; kk < K; ++kk) gh = fmaf(N.head_w[kk * H + k], sh_gl[kk], gh);
      const float s4 = sh_part[0][k] + sh_part[1][k] + sh_part[2][k] + sh_part[3][k];   // W_hh^T d_pre of step t+1
      const float ig = W.gates[q * GH + k], fg = W.gates[q * GH + H + k], gg = W.gates[q * GH + 2 * H + k],
                  og = W.gates[q * GH + 3 * H + k];
      const float ct = W.c_seq[q * H + k];
      const float cprev = t > 0 ? W.c_seq[(q - 1) * H + k] : 0.0f;
      const float tc = w_tanh(ct);
      const float dh = gh + s4;
      const float dct = dh * og * (1.0f - tc * tc) + dc;
      const float dai = dct * gg * ig * (1.0f - ig);
      const float daf = dct * cprev * fg * (1.0f - fg);
      const float dag = dct * ig * (1.0f - gg * gg);
      const float dao = dh * tc * og * (1.0f - og);
      dc = dct * fg;
      sh_d[0][k] = dai; sh_d[1][k] = daf; sh_d[2][k] = dag; sh_d[3][k] = dao;
      W.d_pre[q * GH + k] = dai; W.d_pre[q * GH + H + k] = daf; W.d_pre[q * GH + 2 * H + k] = dag; W.d_pre[q * GH + 3 * H + k] = dao;
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
    sh_part[g][k] = acc;     // k < H: W_hh^T d_pre (dh for step t-1); H <= k < H+I: W_ih^T d_pre = gx_t
    __syncthreads();
  }
  filter_bwd_end(d, G, b, L);
}

extern "C" int kvae_wide_launch_alpha_lstm_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved,
                                               const kvae_lgssm_states *up, const kvae_lgssm_input_grads *out, float *ws, int with_rts,
                                               const float *w_ih, const float *w_hh, const float *head_w, const float *A,
                                               const float *Bm, const float *C, int K, int H, const float *alpha, const float *gates,
                                               const float *c_seq, const float *g_record_up, const float *g_alpha_up, float *g_record,
                                               float *d_pre, float *g_logit, hipStream_t s) {
  if (H != 50 || p->p != 2 || K < 1 || K > 16) return KVAE_ERR_DIMS;
  const AlphaNet net{w_ih, w_hh, nullptr, nullptr, head_w, nullptr, A, Bm, C, K};
  const AlphaBwd w{alpha, gates, c_seq, g_record_up, g_alpha_up, g_record, d_pre, g_logit};
  if (p->n == 4 && p->m == 4)
    k_alpha_lstm_bwd<SDims<4, 4, 2>, 50, 2><<<dim3(p->B), dim3(256), 0, s>>>(*p, *saved, *up, *out, ws, with_rts, net, w);
  else
    k_alpha_lstm_bwd<RDims, 50, 2><<<dim3(p->B), dim3(256), 0, s>>>(*p, *saved, *up, *out, ws, with_rts, net, w);
  return KVAE_OK;
}
