// kvae_lgssm_tpp.hip — the ELBO bodies (lgssm_elbo.h) built with ONE THREAD per (sequence, step) instead of one
// wavefront: the ELBO has no recursion in t, a 4x4 problem keeps at most 16 of a wavefront's 64 lanes busy and runs its
// triangular solves on lane 0, so at large batch the wave-per-step kernel leaves the chip idle (B = 32768, T = 50:
// 3.7 ms = 126 GB/s algorithmic).  Here every lane owns a whole (b,t); the phase structure of the body collapses to
// straight-line code (KV_PAR = serial loop, KV_SYNC = nothing, the ElboLds struct is thread-private), exactly the
// form the host simulation checks against the oracle.  Separate translation unit (own namespace) because the
// execution model is a compile-time property of the bodies.
#define KV_TPP 1
#define kvae kvae_tpp
#include <hip/hip_runtime.h>

#include "lgssm_elbo.h"
#include "regime.h"

using namespace kvae;

template <class D>
__global__ __launch_bounds__(64) void k_elbo_probe_tpp(kvae_lgssm_problem P, const float *Sig_s, const float *mus,
                                                       const float *eps, float *ws, int32_t *levels) {
  const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (q >= (int64_t)P.B * P.T) return;
  ElboLds<D> L;
  const D d(P.n, P.m, P.p);
  const int b = (int)(q / P.T), t = (int)(q - (int64_t)b * P.T);
  elbo_probe_body(d, P, Sig_s, mus, eps, ws, levels, b, t, L);
}

template <class D>
__global__ __launch_bounds__(64) void k_elbo_tpp(kvae_lgssm_problem P, const float *mus, const float *Sigs, const float *eps,
                                                 float *terms, const int32_t *levels, const float *ws, float *g_mus,
                                                 float *g_Sigs, kvae_lgssm_input_grads G, int have_g) {
  const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (q == 0) const_cast<int32_t *>(levels)[2] = 1;   // kernel family of this launch (include/kvae_lgssm.h: chol_levels[2])
  if (q >= (int64_t)P.B * P.T) return;
  ElboLds<D> L;
  const D d(P.n, P.m, P.p);
  const int b = (int)(q / P.T), t = (int)(q - (int64_t)b * P.T);
  elbo_body(d, P, mus, Sigs, eps, terms, levels, ws, g_mus, g_Sigs, have_g ? &G : nullptr, b, t, L);
}

// launchers used by kvae_lgssm.hip (not part of the public C ABI); n = m = 4, p = 2 only
extern "C" void kvae_tpp_launch_elbo_probe(const kvae_lgssm_problem *p, const float *Sig_s, const float *mus, const float *eps,
                                           float *ws, int32_t *levels, hipStream_t s) {
  const unsigned grid = (unsigned)(((int64_t)p->B * p->T + 63) / 64);
  k_elbo_probe_tpp<SDims<4, 4, 2>><<<dim3(grid), dim3(64), 0, s>>>(*p, Sig_s, mus, eps, ws, levels);
}
extern "C" void kvae_tpp_launch_elbo(const kvae_lgssm_problem *p, const float *mus, const float *Sigs, const float *eps,
                                     float *terms, const int32_t *levels, const float *ws, float *g_mus, float *g_Sigs,
                                     const kvae_lgssm_input_grads *g, int have_g, hipStream_t s) {
  const unsigned grid = (unsigned)(((int64_t)p->B * p->T + 63) / 64);
  k_elbo_tpp<SDims<4, 4, 2>><<<dim3(grid), dim3(64), 0, s>>>(*p, mus, Sigs, eps, terms, levels, ws, g_mus, g_Sigs, *g, have_g);
}

// ---------------------------------------------------------------------------------------------------------------
// Regime chain (regime.h) with one thread per SEQUENCE and the regime count K a compile-time constant: the K x K work
// of a step is a few dozen flops, so a wavefront per sequence spends its time in LDS round trips and wave syncs
// (119 us forward / 167 us backward at B = 256, T = 50, K = 3); a single lane walks the T steps in registers.
// ---------------------------------------------------------------------------------------------------------------
template <int KC>
__global__ __launch_bounds__(64) void k_regime_fwd_tpp(const float *logits, const float *init_logits, const float *gumbel,
                                                       const float *P, float *y_seq, float *log_q, float *log_p, int B, int T,
                                                       float tau, const float *tau_dev, int hard) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  if (tau_dev) tau = *tau_dev;
  RegimeLds L;
  regime_fwd_body(logits, init_logits, gumbel, P, y_seq, log_q, log_p, b, T, KC, tau, hard, L);
}
template <int KC>
__global__ __launch_bounds__(64) void k_regime_bwd_tpp(const float *logits, const float *init_logits, const float *gumbel,
                                                       const float *P, const float *y_seq, const float *g_y, const float *g_lq,
                                                       const float *g_lp, float *g_logits, float *g_init, int B, int T, float tau,
                                                       const float *tau_dev) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  if (tau_dev) tau = *tau_dev;
  RegimeLds L;
  regime_bwd_body(logits, init_logits, gumbel, P, y_seq, g_y, g_lq, g_lp, g_logits, g_init, b, T, KC, tau, L);
}

#define KVAE_REGIME_TPP_CASES(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
// return 1 if a thread-per-sequence instance exists for K (and was launched), 0 otherwise
extern "C" int kvae_tpp_launch_regime_fwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                          float *y_seq, float *log_q, float *log_p, int B, int T, int K, float tau,
                                          const float *tau_dev, int hard, hipStream_t s) {
  const dim3 grid((unsigned)((B + 63) / 64));
  switch (K) {
#define X(KC) case KC: k_regime_fwd_tpp<KC><<<grid, dim3(64), 0, s>>>(logits, init_logits, gumbel, P, y_seq, log_q, log_p, B, T, tau, tau_dev, hard); return 1;
    KVAE_REGIME_TPP_CASES(X)
#undef X
    default: return 0;
  }
}
extern "C" int kvae_tpp_launch_regime_bwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                          const float *y_seq, const float *g_y, const float *g_lq, const float *g_lp,
                                          float *g_logits, float *g_init, int B, int T, int K, float tau, const float *tau_dev,
                                          hipStream_t s) {
  const dim3 grid((unsigned)((B + 63) / 64));
  switch (K) {
#define X(KC) case KC: k_regime_bwd_tpp<KC><<<grid, dim3(64), 0, s>>>(logits, init_logits, gumbel, P, y_seq, g_y, g_lq, g_lp, g_logits, g_init, B, T, tau, tau_dev); return 1;
    KVAE_REGIME_TPP_CASES(X)
#undef X
    default: return 0;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Regime chain, one wavefront per sequence as an 8 x 8 lane grid (regime_grid.h): K <= 8, the latency-optimal layout while
// the batch leaves wave slots free
// ---------------------------------------------------------------------------------------------------------------
#include "regime_grid.h"

__global__ __launch_bounds__(64) void k_regime_fwd_grid(const float *logits, const float *init_logits, const float *gumbel,
                                                        const float *P, float *y_seq, float *log_q, float *log_p, int T, int K,
                                                        float tau, const float *tau_dev, int hard) {
  if (tau_dev) tau = *tau_dev;
  kvae::rgrid::regime_fwd(logits, init_logits, gumbel, P, y_seq, log_q, log_p, blockIdx.x, T, K, tau, hard);
}
__global__ __launch_bounds__(64) void k_regime_bwd_grid(const float *logits, const float *init_logits, const float *gumbel,
                                                        const float *P, const float *y_seq, const float *g_y, const float *g_lq,
                                                        const float *g_lp, float *g_logits, float *g_init, int T, int K, float tau,
                                                        const float *tau_dev) {
  if (tau_dev) tau = *tau_dev;
  kvae::rgrid::regime_bwd(logits, init_logits, gumbel, P, y_seq, g_y, g_lq, g_lp, g_logits, g_init, blockIdx.x, T, K, tau);
}
// 1 if launched.  Above ~4096 sequences the chip's wave slots are full and 64 sequences per wavefront (thread-per-sequence)
// have the better throughput.
extern "C" int kvae_grid_launch_regime_fwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                           float *y_seq, float *log_q, float *log_p, int B, int T, int K, float tau,
                                           const float *tau_dev, int hard, hipStream_t s) {
  if (K > 8 || B > 4096) return 0;
  k_regime_fwd_grid<<<dim3(B), dim3(64), 0, s>>>(logits, init_logits, gumbel, P, y_seq, log_q, log_p, T, K, tau, tau_dev, hard);
  return 1;
}
extern "C" int kvae_grid_launch_regime_bwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                           const float *y_seq, const float *g_y, const float *g_lq, const float *g_lp,
                                           float *g_logits, float *g_init, int B, int T, int K, float tau, const float *tau_dev,
                                           hipStream_t s) {
  if (K > 8 || B > 4096) return 0;
  k_regime_bwd_grid<<<dim3(B), dim3(64), 0, s>>>(logits, init_logits, gumbel, P, y_seq, g_y, g_lq, g_lp, g_logits, g_init, T, K, tau,
                                                 tau_dev);
  return 1;
}
