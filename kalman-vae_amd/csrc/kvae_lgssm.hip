// kvae_lgssm.hip — gfx950 kernels and the C ABI (include/kvae_lgssm.h) of the LGSSM hot path.
//
// Launch geometry: every workgroup is ONE 64-lane wavefront.
//   smooth / filter / rts / backward : grid = B   (one wavefront per sequence, T loop in-kernel)
//   elbo probe / elbo                : grid = B*T (one wavefront per (sequence, step))
//   mix                              : 256-thread element-wise / slab-reduction kernels
// Dimensions (n,m,p) = (4,4,2) and (16,16,2) get fully unrolled compile-time specialisations;
// anything else (<= 16) runs the run-time-dimension instantiation of the same bodies.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lgssm_bwd.h"
#include "lgssm_elbo.h"
#include "lgssm_fwd.h"
#include "lgssm_n4.h"
#include "mix.h"

using namespace kvae;

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
template <class D>
__global__ __launch_bounds__(64) void k_smooth_fwd(kvae_lgssm_problem P, kvae_lgssm_states S, int do_filter,
                                                   int do_rts) {
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
__global__ __launch_bounds__(64) void k_smooth_bwd(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
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

// n = 4, p = 2 fused-phase kernels (lgssm_n4.h)
template <class D>
__global__ __launch_bounds__(64) void k_smooth_fwd_n4(kvae_lgssm_problem P, kvae_lgssm_states S, int do_filter, int do_rts) {
  __shared__ N4Lds<D::MMAX> L;
  const D d(P.n, P.m, P.p);
  const int b = blockIdx.x;
  if (do_filter) {
    filter_sweep_n4(d, P, S, b, L);
    KV_SYNC();
  }
  if (do_rts) rts_sweep_n4(d, P, S, b, L);
}

template <class D>
__global__ __launch_bounds__(64) void k_smooth_bwd_n4(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
                                                      kvae_lgssm_input_grads G, float *ws, int with_rts) {
  __shared__ N4BwdLds<D::MMAX> L;
  const D d(P.n, P.m, P.p);
  const int b = blockIdx.x;
  if (with_rts)
    rts_bwd_sweep_n4(d, P, S, U, G, ws, b, L);
  else
    filter_bwd_seed(d, P, U, G, ws, b);
  KV_SYNC();
  filter_bwd_sweep_n4(d, P, S, G, ws, b, L);
}

template <class D>
__global__ __launch_bounds__(64) void k_elbo_probe(kvae_lgssm_problem P, const float *Sig_s, const float *mus,
                                                   const float *eps, float *ws, int32_t *levels) {
  __shared__ ElboLds<D> L;
  const D d(P.n, P.m, P.p);
  const int b = blockIdx.x / P.T, t = blockIdx.x - b * P.T;
  elbo_probe_body(d, P, Sig_s, mus, eps, ws, levels, b, t, L);
}

template <class D>
__global__ __launch_bounds__(64) void k_elbo(kvae_lgssm_problem P, const float *mus, const float *Sigs,
                                             const float *eps, float *terms, const int32_t *levels, const float *ws,
                                             float *g_mus, float *g_Sigs, kvae_lgssm_input_grads G, int have_g,
                                             int only_if_jitter) {
  // only_if_jitter: this launch backs up a level-0-only fast kernel (lgssm_n16_elbo.h) that has already run
  if (only_if_jitter && levels[0] == 0 && levels[1] == 0) return;
  __shared__ ElboLds<D> L;
  const D d(P.n, P.m, P.p);
  // grid-stride over the (b, t) cells: the backup launch is a few thousand workgroups that normally leave at the line above
  for (int64_t w = blockIdx.x; w < (int64_t)P.B * P.T; w += gridDim.x) {
    const int b = (int)(w / P.T), t = (int)(w - (int64_t)b * P.T);
    elbo_body(d, P, mus, Sigs, eps, terms, levels, ws, g_mus, g_Sigs, have_g ? &G : nullptr, b, t, L);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_mix_fwd(const float *alpha, const float *base, float *out, int64_t total,
                                                 int K, int E) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) mix_fwd_elem(alpha, base, out, idx, K, E);
}

__global__ __launch_bounds__(256) void k_mix_bwd_alpha(const float *base, const float *g_out, float *g_alpha,
                                                       int64_t total, int K, int E, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx < total) mix_bwd_alpha_elem(base, g_out, g_alpha, idx, K, E, accumulate);
}

// Wide records (E >= 128, i.e. n = 16: E = 544 / 768): one wavefront per row, lanes stride over E with coalesced reads of
// g_out and base, K running sums per lane, butterfly reduction.  The thread-per-(row,k) kernel above walks E serially from
// every thread (fine for E = 40, 6 us at configs[1]; 528 us at the configs[4] shard, where this one takes ~60).
template <int KC>
__global__ __launch_bounds__(256) void k_mix_bwd_alpha_wide(const float *__restrict__ base, const float *__restrict__ g_out,
                                                            float *__restrict__ g_alpha, int64_t rows, int K, int E, int accumulate) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= rows) return;
  float acc[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) acc[k] = 0.f;
  for (int e = lane; e < E; e += 64) {
    const float g = g_out[r * E + e];
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k < K) acc[k] = fmaf(g, base[k * E + e], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    float v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0 && k < K) g_alpha[r * K + k] = accumulate ? g_alpha[r * K + k] + v : v;
  }
}

__global__ void k_mix_bwd_partial(const float *alpha, const float *g_out, float *partials, int64_t rows, int K, int E) {
  const int e = blockIdx.y * blockDim.x + threadIdx.x;
  if (e < E) mix_bwd_partial_elem(alpha, g_out, partials, blockIdx.x, e, rows, K, E);
}

// one wavefront per output element: lanes stride over the slab partials, then a fixed-order butterfly
// (deterministic: the same summation tree on every run)
__global__ __launch_bounds__(64) void k_mix_bwd_final(const float *partials, float *g_base, int64_t nblk, int KE) {
  const int idx = blockIdx.x;
  float acc = 0.f;
  for (int64_t blk = threadIdx.x; blk < nblk; blk += 64) acc += partials[blk * KE + idx];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) g_base[idx] = acc;
}

#include "mix_wave.h"

// ---------------------------------------------------------------------------------------------
// host side: validation, dispatch on (n,m,p), launch
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";

static int launch_status(const char *what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return KVAE_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return KVAE_ERR_LAUNCH;
}

static int check_problem(const kvae_lgssm_problem *p) {
  if (!p) return KVAE_ERR_NULL;
  if (p->B < 1 || p->T < 1 || p->n < 1 || p->m < 1 || p->p < 1 || p->n > KVAE_MAX_DIM || p->m > KVAE_MAX_DIM ||
      p->p > KVAE_MAX_DIM)
    return KVAE_ERR_DIMS;
  if (!p->A.ptr || !p->Bm.ptr || !p->C.ptr || !p->Q.ptr || !p->R || !p->mu0 || !p->Sigma0 || !p->Y || !p->U)
    return KVAE_ERR_NULL;
  return KVAE_OK;
}

#define KVAE_DISPATCH(P, ...)                                    \
  do {                                                           \
    if ((P).n == 4 && (P).m == 4 && (P).p == 2) {               \
      using D = SDims<4, 4, 2>;                                  \
      __VA_ARGS__;                                                   \
    } else if ((P).n == 16 && (P).m == 16 && (P).p == 2) {      \
      using D = SDims<16, 16, 2>;                                \
      __VA_ARGS__;                                                   \
    } else {                                                     \
      using D = RDims;                                           \
      __VA_ARGS__;                                                   \
    }                                                            \
  } while (0)

// kvae_lgssm_wide.hip: the same bodies with 256 threads per sequence (used when n > 8)
#define KVAE_ELBO_TPP_MIN_STEPS 0   /* thread-per-step wins at every size measured (12.8k .. 1.6M steps), DESIGN.md */
extern "C" void kvae_tpp_launch_elbo_probe(const kvae_lgssm_problem *p, const float *Sig_s, const float *mus, const float *eps,
                                           float *ws, int32_t *levels, hipStream_t s);
extern "C" void kvae_tpp_launch_elbo(const kvae_lgssm_problem *p, const float *mus, const float *Sigs, const float *eps,
                                     float *terms, const int32_t *levels, const float *ws, float *g_mus, float *g_Sigs,
                                     const kvae_lgssm_input_grads *g, int have_g, hipStream_t s);
extern "C" int kvae_tpp_launch_regime_fwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                          float *y_seq, float *log_q, float *log_p, int B, int T, int K, float tau,
                                          const float *tau_dev, int hard, hipStream_t s);
extern "C" int kvae_tpp_launch_regime_bwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                          const float *y_seq, const float *g_y, const float *g_lq, const float *g_lp,
                                          float *g_logits, float *g_init, int B, int T, int K, float tau, const float *tau_dev,
                                          hipStream_t s);
extern "C" int kvae_grid_launch_regime_fwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                           float *y_seq, float *log_q, float *log_p, int B, int T, int K, float tau,
                                           const float *tau_dev, int hard, hipStream_t s);
extern "C" int kvae_grid_launch_regime_bwd(const float *logits, const float *init_logits, const float *gumbel, const float *P,
                                           const float *y_seq, const float *g_y, const float *g_lq, const float *g_lp,
                                           float *g_logits, float *g_init, int B, int T, int K, float tau, const float *tau_dev,
                                           hipStream_t s);
extern "C" void kvae_wide_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                     hipStream_t s);
// kvae_lgssm_n16.hip: (n, m, p) = (16, 16, 2) on the f32 matrix cores
extern "C" void kvae_n16_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                    hipStream_t s);

extern "C" void kvae_n16_launch_elbo_probe(const kvae_lgssm_problem *p, const float *Sig_s, const float *mus, const float *eps,
                                           float *zst, int32_t *levels, hipStream_t s);
extern "C" void kvae_n16_launch_elbo(const kvae_lgssm_problem *p, const float *mus, const float *Sigs, const float *eps,
                                     float *terms, const int32_t *levels, const float *zst, float *g_mus, float *g_Sigs,
                                     const kvae_lgssm_input_grads *g, int have_g, hipStream_t s);
// kvae_lgssm_n16.hip (lgssm_m4.h over the quad layout of lgssm_q4.h): (n, m, p) = (4, 4, 2), sixteen sequences per wavefront
extern "C" void kvae_q4_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                   hipStream_t s);
extern "C" void kvae_q4_launch_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                                   const kvae_lgssm_input_grads *out, float *ws, int has_fp, hipStream_t s);
extern "C" void kvae_n16_launch_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                                    const kvae_lgssm_input_grads *out, float *ws, int has_fp, hipStream_t s);

// The n = 16 kernels move matrices with 16-byte accesses: every per-step operand must start on a 16-byte boundary.
static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static bool stack16(const kvae_stack &s) { return aligned16(s.ptr) && s.sb % 4 == 0 && s.st % 4 == 0; }
static bool q4_ok(const kvae_lgssm_problem *p) {
  static const int env = getenv("KVAE_Q4") ? atoi(getenv("KVAE_Q4")) : 1;   // 0: one wavefront per sequence (A/B runs)
  // lgssm_m4.h reads A, B, Q and the two rows of C as 16-byte rows, and (the adjoint) u_t and the previous mean whole
  return env && p->n == 4 && p->m == 4 && p->p == 2 && stack16(p->A) && stack16(p->Bm) && stack16(p->Q) && stack16(p->C) &&
         aligned16(p->Sigma0) && p->Sigma0_sb % 4 == 0 && aligned16(p->mu0) && p->mu0_sb % 4 == 0 && aligned16(p->U) &&
         (reinterpret_cast<uintptr_t>(p->Y) & 7u) == 0;
}
static bool n16_ok(const kvae_lgssm_problem *p) {
  static const int env = getenv("KVAE_N16") ? atoi(getenv("KVAE_N16")) : 1;   // 0: generic kernels (A/B runs)
  return env && p->n == 16 && p->m == 16 && p->p == 2 && stack16(p->A) && stack16(p->Bm) && stack16(p->C) && stack16(p->Q) &&
         aligned16(p->mu0) && p->mu0_sb % 4 == 0 && aligned16(p->Sigma0) && p->Sigma0_sb % 4 == 0 && aligned16(p->U);
}
extern "C" int kvae_wide_launch_filter_alpha_lstm(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, const float *w_ih,
                                                  const float *w_hh, const float *b_ih, const float *b_hh, const float *head_w,
                                                  const float *head_b, const float *A, const float *Bm, const float *C, int K,
                                                  int H, float *record, float *alpha, float *gates, float *c_seq, float *h_seq,
                                                  float *x_seq, hipStream_t s);
extern "C" int kvae_wide_launch_alpha_lstm_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved,
                                               const kvae_lgssm_states *up, const kvae_lgssm_input_grads *out, float *ws, int with_rts,
                                               const float *w_ih, const float *w_hh, const float *head_w, const float *A,
                                               const float *Bm, const float *C, int K, int H, const float *alpha, const float *gates,
                                               const float *c_seq, const float *g_record_up, const float *g_alpha_up, float *g_record,
                                               float *d_pre, float *g_logit, hipStream_t s);

static int launch_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *st, int do_filter, int do_rts,
                      void *stream) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!st || !st->mus_filt || !st->Sigmas_filt || !st->mus_pred || !st->Sigmas_pred) return KVAE_ERR_NULL;
  if (do_rts && (!st->mus_smooth || !st->Sigmas_smooth)) return KVAE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (q4_ok(prob) && aligned16(st->Sigmas_filt) && aligned16(st->Sigmas_pred) && aligned16(st->Sigmas_smooth) &&
      aligned16(st->aux)) {
    kvae_q4_launch_fwd(prob, st, do_filter, do_rts, s);
    return launch_status("k_smooth_fwd_m4");
  }
  if (prob->n == 4 && prob->m == 4 && prob->p == 2 && (st->aux || !do_filter)) {
    // rts-only calls need no gains; filter calls use the fused-phase kernel when the caller provides aux
    k_smooth_fwd_n4<SDims<4, 4, 2>><<<dim3(prob->B), dim3(64), 0, s>>>(*prob, *st, do_filter, do_rts);
    return launch_status("k_smooth_fwd_n4");
  }
  if (n16_ok(prob) && aligned16(st->mus_filt) && aligned16(st->Sigmas_filt) && aligned16(st->mus_pred) &&
      aligned16(st->Sigmas_pred) && aligned16(st->mus_smooth) && aligned16(st->Sigmas_smooth) && aligned16(st->aux)) {
    kvae_n16_launch_fwd(prob, st, do_filter, do_rts, s);
    return launch_status("k_smooth_fwd_n16");
  }
  if (prob->n > 8) {
    kvae_wide_launch_fwd(prob, st, do_filter, do_rts, s);
    return launch_status("k_smooth_fwd_wide");
  }
  KVAE_DISPATCH(*prob, k_smooth_fwd<D><<<dim3(prob->B), dim3(64), 0, s>>>(*prob, *st, do_filter, do_rts));
  return launch_status("k_smooth_fwd");
}

// the streaming mixture kernels (mix_wave.h): 16-byte pieces, K base records in registers
static bool mix_wave_ok(const float *a, const float *b, const float *c, int K, int E) {
  // (E >= 128: with a record of 40 floats - n = 4 - a wavefront per row leaves 54 lanes idle and the element-wise kernels win)
  return (E & 3) == 0 && E >= 128 && E <= 768 && K <= 8 && aligned16(b) && aligned16(c) && a != nullptr;
}
static int64_t mix_wave_slabs(int64_t rows, int64_t cap) {   // at least 8 rows per wavefront
  const int64_t want = (rows + 7) / 8;
  return want < 1 ? 1 : (want > cap ? cap : want);
}
#define KVAE_MIXW_DISPATCH(L)                        \
  do {                                               \
    const int ej = (E + 255) / 256;                  \
    if (K <= 4) {                                    \
      if (ej == 1) L(4, 1);                          \
      else if (ej == 2) L(4, 2);                     \
      else L(4, 3);                                  \
    } else {                                         \
      if (ej == 1) L(8, 1);                          \
      else if (ej == 2) L(8, 2);                     \
      else L(8, 3);                                  \
    }                                                \
  } while (0)

extern "C" {

int kvae_lgssm_filter_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, void *stream) {
  return launch_fwd(prob, out, 1, 0, stream);
}
int kvae_lgssm_rts_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *io, void *stream) {
  return launch_fwd(prob, io, 0, 1, stream);
}
int kvae_lgssm_smooth_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, void *stream) {
  return launch_fwd(prob, out, 1, 1, stream);
}

int kvae_lgssm_filter_alpha_lstm(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, const float *w_ih,
                                 const float *w_hh, const float *b_ih, const float *b_hh, const float *head_w,
                                 const float *head_b, const float *A, const float *Bm, const float *C, int32_t K, int32_t H,
                                 float *record, float *alpha, float *gates, float *c_seq, float *h_seq, float *x_seq,
                                 void *stream) {
  if (!prob || !out) return KVAE_ERR_NULL;
  if (prob->B < 1 || prob->T < 1 || prob->n < 1 || prob->m < 1 || prob->n > KVAE_MAX_DIM || prob->m > KVAE_MAX_DIM)
    return KVAE_ERR_DIMS;
  if (!prob->Q.ptr || !prob->R || !prob->mu0 || !prob->Sigma0 || !prob->Y || !prob->U || !w_ih || !w_hh || !b_ih || !b_hh ||
      !head_w || !head_b || !A || !Bm || !C || !record || !alpha || !out->mus_filt || !out->Sigmas_filt || !out->mus_pred ||
      !out->Sigmas_pred)
    return KVAE_ERR_NULL;
  const int rc = kvae_wide_launch_filter_alpha_lstm(prob, out, w_ih, w_hh, b_ih, b_hh, head_w, head_b, A, Bm, C, K, H, record,
                                                    alpha, gates, c_seq, h_seq, x_seq, (hipStream_t)stream);
  return rc ? rc : launch_status("k_filter_alpha_lstm");
}

int kvae_lgssm_alpha_lstm_bwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                              const kvae_lgssm_input_grads *out, float *ws, int with_rts, const float *w_ih,
                              const float *w_hh, const float *head_w, const float *A, const float *Bm, const float *C,
                              int32_t K, int32_t H, const float *alpha, const float *gates, const float *c_seq,
                              const float *g_record_up, const float *g_alpha_up, float *g_record, float *d_pre,
                              float *g_logit, void *stream) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!saved || !up || !out || !ws || !w_ih || !w_hh || !head_w || !A || !Bm || !C || !alpha || !gates || !c_seq || !g_record ||
      !d_pre || !g_logit)
    return KVAE_ERR_NULL;
  if (!saved->mus_filt || !saved->Sigmas_filt || !saved->mus_pred || !saved->Sigmas_pred) return KVAE_ERR_NULL;
  if (with_rts && (!saved->mus_smooth || !saved->Sigmas_smooth)) return KVAE_ERR_NULL;
  if (!out->gA.ptr || !out->gB.ptr || !out->gC.ptr || !out->gY) return KVAE_ERR_NULL;
  rc = kvae_wide_launch_alpha_lstm_bwd(prob, saved, up, out, ws, with_rts, w_ih, w_hh, head_w, A, Bm, C, K, H, alpha, gates, c_seq,
                                       g_record_up, g_alpha_up, g_record, d_pre, g_logit, (hipStream_t)stream);
  return rc ? rc : launch_status("k_alpha_lstm_bwd");
}

int kvae_lgssm_smooth_bwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                          const kvae_lgssm_input_grads *out, float *ws, int with_rts, void *stream) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!saved || !up || !out || !ws) return KVAE_ERR_NULL;
  if (!saved->mus_filt || !saved->Sigmas_filt || !saved->mus_pred || !saved->Sigmas_pred) return KVAE_ERR_NULL;
  if (with_rts && (!saved->mus_smooth || !saved->Sigmas_smooth)) return KVAE_ERR_NULL;
  if (!out->gA.ptr || !out->gB.ptr || !out->gC.ptr || !out->gY) return KVAE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (with_rts && saved->aux && q4_ok(prob)) {   // sixteen sequences per wavefront (lgssm_m4.h)
    const int fp = (up->mus_filt != nullptr) + (up->Sigmas_filt != nullptr) + (up->mus_pred != nullptr) + (up->Sigmas_pred != nullptr);
    const auto gs16 = [](const kvae_gstack &g) { return !g.ptr || (aligned16(g.ptr) && g.sb % 4 == 0 && g.st % 4 == 0); };
    const bool al = aligned16(saved->mus_filt) && aligned16(saved->Sigmas_filt) && aligned16(saved->Sigmas_pred) &&
                    aligned16(saved->Sigmas_smooth) && aligned16(saved->aux) && aligned16(ws) && aligned16(up->Sigmas_smooth) &&
                    aligned16(up->Sigmas_filt) && aligned16(up->Sigmas_pred) && gs16(out->gA) && gs16(out->gB) && gs16(out->gQ) &&
                    aligned16(out->g_Sigma0);
    if (up->mus_smooth && up->Sigmas_smooth && (fp == 0 || fp == 4) && al && out->gU) {
      kvae_q4_launch_bwd(prob, saved, up, out, ws, fp == 4, s);
      return launch_status("k_smooth_bwd_m4");
    }
  }
  if (prob->n == 4 && prob->m == 4 && prob->p == 2 && saved->aux) {
    k_smooth_bwd_n4<SDims<4, 4, 2>><<<dim3(prob->B), dim3(64), 0, s>>>(*prob, *saved, *up, *out, ws, with_rts);
    return launch_status("k_smooth_bwd_n4");
  }
  if (with_rts && saved->aux && n16_ok(prob)) {
    // matrix-core backward (lgssm_n16.h): needs the gains saved by the forward, the upstream gradient of the smoothed
    // stacks, and the upstream gradients of the filtered / predicted stacks either all present or all absent
    const int fp = (up->mus_filt != nullptr) + (up->Sigmas_filt != nullptr) + (up->mus_pred != nullptr) + (up->Sigmas_pred != nullptr);
    const bool al = aligned16(saved->mus_filt) && aligned16(saved->Sigmas_filt) && aligned16(saved->mus_pred) &&
                    aligned16(saved->Sigmas_pred) && aligned16(saved->mus_smooth) && aligned16(saved->Sigmas_smooth) &&
                    aligned16(saved->aux) && aligned16(ws) && aligned16(up->Sigmas_smooth) && aligned16(up->Sigmas_filt) &&
                    aligned16(up->Sigmas_pred) && aligned16(out->gA.ptr) && out->gA.sb % 4 == 0 && out->gA.st % 4 == 0;
    if (up->mus_smooth && up->Sigmas_smooth && (fp == 0 || fp == 4) && al && out->gU) {
      kvae_n16_launch_bwd(prob, saved, up, out, ws, fp == 4, s);
      return launch_status("k_smooth_bwd_n16");
    }
  }
  KVAE_DISPATCH(*prob, k_smooth_bwd<D><<<dim3(prob->B), dim3(64), 0, s>>>(*prob, *saved, *up, *out, ws,
                                           with_rts));
  return launch_status("k_smooth_bwd");
}

int kvae_lgssm_elbo(const kvae_lgssm_problem *prob, const float *mus_smooth, const float *Sigmas_smooth, const float *eps,
                    float *terms, int32_t *chol_levels, float *ws_lz, float *g_mus, float *g_Sigmas,
                    const kvae_lgssm_input_grads *g, void *stream) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!mus_smooth || !Sigmas_smooth || !eps || !terms || !chol_levels) return KVAE_ERR_NULL;
  const bool want_g = (g_mus != nullptr);
  if (want_g && (!g_Sigmas || !g || !g->gA.ptr || !g->gB.ptr || !g->gC.ptr || !g->gY)) return KVAE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(chol_levels, 0, 3 * sizeof(int32_t), s) != hipSuccess) return launch_status("memset chol_levels");
  const unsigned grid = (unsigned)((int64_t)prob->B * prob->T);
  kvae_lgssm_input_grads gz;
  memset(&gz, 0, sizeof(gz));
  // n = 4: one THREAD per (b,t) (kvae_lgssm_tpp.hip); KVAE_ELBO_TPP=0 selects the wave-per-step kernels for A/B runs
  static const int tpp_env = getenv("KVAE_ELBO_TPP") ? atoi(getenv("KVAE_ELBO_TPP")) : -1;
  const bool n4 = prob->n == 4 && prob->m == 4 && prob->p == 2;
  if (n4 && (tpp_env == 1 || (tpp_env < 0 && (int64_t)prob->B * prob->T >= KVAE_ELBO_TPP_MIN_STEPS))) {
    kvae_tpp_launch_elbo_probe(prob, Sigmas_smooth, mus_smooth, eps, ws_lz, (int32_t *)chol_levels, s);
    rc = launch_status("k_elbo_probe_tpp");
    if (rc) return rc;
    kvae_tpp_launch_elbo(prob, mus_smooth, Sigmas_smooth, eps, terms, (const int32_t *)chol_levels, (const float *)ws_lz, g_mus,
                         g_Sigmas, want_g ? g : &gz, want_g ? 1 : 0, s);
    return launch_status("k_elbo_tpp");
  }
  if (n16_ok(prob) && ws_lz && aligned16(mus_smooth) && aligned16(Sigmas_smooth) && aligned16(eps) && aligned16(ws_lz) &&
      (!want_g || (aligned16(g_Sigmas) && aligned16(g->gA.ptr) && g->gA.sb % 4 == 0 && g->gA.st % 4 == 0 && aligned16(g->gB.ptr) &&
                   g->gB.sb % 4 == 0 && g->gB.st % 4 == 0))) {
    // matrix-core / four-matrices-per-wavefront kernels (lgssm_n16_elbo.h): the probe resolves the jitter levels with the
    // generic semantics (and re-samples z at a raised level of Sigma_s); the main launch computes the call at whatever the
    // levels are - jitter ladder and diagonal fallback included, no generic backup launch
    kvae_n16_launch_elbo_probe(prob, Sigmas_smooth, mus_smooth, eps, ws_lz, chol_levels, s);
    rc = launch_status("k_elbo_probe_n16");
    if (rc) return rc;
    kvae_n16_launch_elbo(prob, mus_smooth, Sigmas_smooth, eps, terms, chol_levels, ws_lz, g_mus, g_Sigmas, want_g ? g : &gz,
                         want_g ? 1 : 0, s);
    return launch_status("k_elbo_n16");
  }
  KVAE_DISPATCH(*prob, k_elbo_probe<D><<<dim3(grid), dim3(64), 0, s>>>(*prob, Sigmas_smooth, mus_smooth, eps, ws_lz, chol_levels));
  rc = launch_status("k_elbo_probe");
  if (rc) return rc;
  KVAE_DISPATCH(*prob, k_elbo<D><<<dim3(grid), dim3(64), 0, s>>>(*prob, mus_smooth, Sigmas_smooth, eps,
                                           terms, (const int32_t *)chol_levels, (const float *)ws_lz, g_mus, g_Sigmas, want_g ? *g : gz,
                                           want_g ? 1 : 0, 0));
  return launch_status("k_elbo");
}

int kvae_mix_fwd(const float *alpha, const float *base, float *out, int64_t rows, int32_t K, int32_t E, void *stream) {
  if (!alpha || !base || !out) return KVAE_ERR_NULL;
  if (rows < 1 || K < 1 || K > KVAE_MAX_K || E < 1) return KVAE_ERR_ARG;
  const int64_t total = rows * E;
  if (mix_wave_ok(alpha, base, out, K, E)) {   // streaming kernels (mix_wave.h)
    const int64_t slabs = mix_wave_slabs(rows, 4096), per = (rows + slabs - 1) / slabs;   // about four wavefronts per SIMD
    const dim3 grid((unsigned)((slabs + 3) / 4)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define KVAE_MIXW_FWD(KM, EJ) mixw::k_mix_fwd_wave<KM, EJ><<<grid, block, 0, s>>>(alpha, base, out, rows, K, E, per)
    KVAE_MIXW_DISPATCH(KVAE_MIXW_FWD);
#undef KVAE_MIXW_FWD
    return launch_status("k_mix_fwd_wave");
  }
  hipLaunchKernelGGL(k_mix_fwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, alpha, base, out,
                     total, K, E);
  return launch_status("k_mix_fwd");
}

int64_t kvae_mix_bwd_partials(int64_t rows) { return (rows + KVAE_MIX_ROWS_PER_BLOCK - 1) / KVAE_MIX_ROWS_PER_BLOCK; }

int kvae_mix_bwd(const float *alpha, const float *base, const float *g_out, float *g_alpha, float *g_base, float *partials,
                 int64_t rows, int32_t K, int32_t E, int32_t accumulate_alpha, void *stream) {
  if (!alpha || !base || !g_out || !g_alpha || !g_base || !partials) return KVAE_ERR_NULL;
  if (rows < 1 || K < 1 || K > KVAE_MAX_K || E < 1) return KVAE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (mix_wave_ok(alpha, base, g_out, K, E) && aligned16(partials)) {   // one pass over g_out (mix_wave.h)
    int64_t slabs = mix_wave_slabs(rows, mixw::MAX_SLABS);              // one register-resident base gradient per wavefront
    const int64_t cap = kvae_mix_bwd_partials(rows);                    // what the caller's partials buffer holds
    slabs = slabs < cap ? slabs : cap;
    const int64_t per = (rows + slabs - 1) / slabs;
    const int nslab = (int)((rows + per - 1) / per);
    const dim3 grid((unsigned)((nslab + 3) / 4)), block(256);
#define KVAE_MIXW_BWD(KM, EJ) \
  mixw::k_mix_bwd_wave<KM, EJ><<<grid, block, 0, s>>>(alpha, base, g_out, g_alpha, partials, rows, K, E, per, nslab, accumulate_alpha)
    KVAE_MIXW_DISPATCH(KVAE_MIXW_BWD);
#undef KVAE_MIXW_BWD
    mixw::k_mix_bwd_fold<<<dim3((unsigned)((K * E + 31) / 32)), block, 0, s>>>(partials, g_base, nslab, K * E);
    return launch_status("k_mix_bwd_wave");
  }
  const int64_t ta = rows * K;
  if (E >= 128 && K <= 4)
    k_mix_bwd_alpha_wide<4><<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(base, g_out, g_alpha, rows, K, E, accumulate_alpha);
  else if (E >= 128)
    k_mix_bwd_alpha_wide<KVAE_MAX_K><<<dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s>>>(base, g_out, g_alpha, rows, K, E,
                                                                                        accumulate_alpha);
  else
    hipLaunchKernelGGL(k_mix_bwd_alpha, dim3((unsigned)((ta + 255) / 256)), dim3(256), 0, s, base, g_out, g_alpha, ta, K, E,
                       accumulate_alpha);
  const int64_t nblk = kvae_mix_bwd_partials(rows);
  const int tpb = E >= 256 ? 256 : ((E + 63) / 64) * 64;
  hipLaunchKernelGGL(k_mix_bwd_partial, dim3((unsigned)nblk, (unsigned)((E + tpb - 1) / tpb)), dim3(tpb), 0, s, alpha, g_out,
                     partials, rows, K, E);
  hipLaunchKernelGGL(k_mix_bwd_final, dim3((unsigned)(K * E)), dim3(64), 0, s, partials, g_base, nblk, K * E);
  return launch_status("k_mix_bwd");
}

int kvae_abi_version(void) { return KVAE_ABI_VERSION; }
const char *kvae_last_error(void) { return g_err; }
#define KVAE_STR2(x) #x
#define KVAE_STR(x) KVAE_STR2(x)
const char *kvae_build_info(void) { return "kvae_lgssm gfx950 (HIP, wave64) abi " KVAE_STR(KVAE_ABI_VERSION); }

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// alpha-network LSTM (lstm.h): one wavefront per sequence, weights in LDS
// ---------------------------------------------------------------------------------------------
#include "lstm.h"
#include "lstm_fast.h"

__global__ __launch_bounds__(64) void k_lstm_fwd(const float *x, const float *w_ih, const float *w_hh, const float *b_ih,
                                                 const float *b_hh, float *h_seq, float *gates, float *c_seq, int T, int I,
                                                 int H) {
  __shared__ LstmLds L;
  lstm_fwd_body(x, w_ih, w_hh, b_ih, b_hh, h_seq, gates, c_seq, blockIdx.x, T, I, H, L);
}

__global__ __launch_bounds__(64) void k_lstm_bwd(const float *g_h, const float *gates, const float *c_seq, const float *w_ih,
                                                 const float *w_hh, float *d_pre, float *dx, int T, int I, int H) {
  __shared__ LstmLds L;
  lstm_bwd_body(g_h, gates, c_seq, w_ih, w_hh, d_pre, dx, blockIdx.x, T, I, H, L);
}

extern "C" {

int kvae_lstm_fwd(const float *x, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, float *h_seq,
                  float *gates, float *c_seq, int32_t B, int32_t T, int32_t I, int32_t H, void *stream) {
  if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !h_seq || !gates || !c_seq) return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || I < 1 || I > KVAE_LSTM_MAX_I || H < 1 || H > KVAE_LSTM_MAX_H) return KVAE_ERR_DIMS;
  if (H == 50 && I == 2)  // KVAEConfig defaults (dynamics_hidden_dim = 50, a_dim = 2): register-resident weights
    k_lstm_fwd_fast<50, 2><<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(x, w_ih, w_hh, b_ih, b_hh, h_seq, gates, c_seq, T);
  else
    k_lstm_fwd<<<dim3(B), dim3(64), 0, (hipStream_t)stream>>>(x, w_ih, w_hh, b_ih, b_hh, h_seq, gates, c_seq, T, I, H);
  return launch_status("k_lstm_fwd");
}

int kvae_lstm_bwd(const float *g_h, const float *gates, const float *c_seq, const float *w_ih, const float *w_hh,
                  float *d_pre, float *dx, int32_t B, int32_t T, int32_t I, int32_t H, void *stream) {
  if (!g_h || !gates || !c_seq || !w_ih || !w_hh || !d_pre || !dx) return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || I < 1 || I > KVAE_LSTM_MAX_I || H < 1 || H > KVAE_LSTM_MAX_H) return KVAE_ERR_DIMS;
  if (H == 50 && I == 2)
    k_lstm_bwd_fast<50, 2><<<dim3(B), dim3(256), 0, (hipStream_t)stream>>>(g_h, gates, c_seq, w_ih, w_hh, d_pre, dx, T);
  else
    k_lstm_bwd<<<dim3(B), dim3(64), 0, (hipStream_t)stream>>>(g_h, gates, c_seq, w_ih, w_hh, d_pre, dx, T, I, H);
  return launch_status("k_lstm_bwd");
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// imputation read-out: a = C_t mu_t for the smoothed and the filtered means in one launch (model.py:279-288)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emission_means(kvae_stack C, const float *__restrict__ ms, const float *__restrict__ mf,
                                                        float *__restrict__ a_s, float *__restrict__ a_f, int64_t BT, int T, int n,
                                                        int p) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= BT * p) return;
  const int64_t q = idx / p;
  const int i = (int)(idx % p);
  const float *c = C.ptr + (q / T) * C.sb + (q % T) * C.st + (int64_t)i * n;
  float s = 0.f, f = 0.f;
  for (int k = 0; k < n; ++k) {
    const float ck = c[k];
    if (ms) s = fmaf(ck, ms[q * n + k], s);
    if (mf) f = fmaf(ck, mf[q * n + k], f);
  }
  if (a_s) a_s[idx] = s;
  if (a_f) a_f[idx] = f;
}
extern "C" int kvae_lgssm_emission_means(const kvae_lgssm_problem *prob, const float *mus_smooth, const float *mus_filt,
                                         float *a_imputed, float *a_filtered, void *stream) {
  if (!prob || !prob->C.ptr || (!mus_smooth != !a_imputed) || (!mus_filt != !a_filtered) || (!a_imputed && !a_filtered))
    return KVAE_ERR_NULL;
  if (prob->B < 1 || prob->T < 1 || prob->n < 1 || prob->p < 1 || prob->n > KVAE_MAX_DIM || prob->p > KVAE_MAX_DIM) return KVAE_ERR_DIMS;
  const int64_t BT = (int64_t)prob->B * prob->T;
  k_emission_means<<<dim3((unsigned)((BT * prob->p + 255) / 256)), dim3(256), 0, (hipStream_t)stream>>>(
      prob->C, mus_smooth, mus_filt, a_imputed, a_filtered, BT, prob->T, prob->n, prob->p);
  return launch_status("k_emission_means");
}

// ---------------------------------------------------------------------------------------------
// alpha-network parameter gradients and linear heads (kvae_lgssm_rnn.hip)
// ---------------------------------------------------------------------------------------------
extern "C" int kvae_rnn_launch_wgrad(const kvae_wgrad_problem *probs, int32_t n, float *ws, hipStream_t s);
extern "C" int kvae_rnn_launch_linear_fwd(const float *x, int64_t xs, int64_t N, int F, const float *W, const float *b, int O,
                                          int softmax, float *y, hipStream_t s);
extern "C" int kvae_rnn_launch_linear_bwd_input(const float *g, const float *y, int64_t N, int F, const float *W, int O,
                                                float *g_logit, float *dx, int64_t dxs, hipStream_t s);
extern "C" {
int kvae_rnn_wgrad(const kvae_wgrad_problem *probs, int32_t n, float *ws, void *stream) {
  if (!probs || !ws) return KVAE_ERR_NULL;
  if (n < 1 || n > 4) return KVAE_ERR_ARG;
  const int rc = kvae_rnn_launch_wgrad(probs, n, ws, (hipStream_t)stream);
  return rc ? rc : launch_status("k_rnn_wgrad");
}
int kvae_linear_fwd(const float *x, int64_t x_stride, int64_t N, int32_t F, const float *W, const float *b, int32_t O,
                    int32_t softmax, float *y, void *stream) {
  if (!x || !W || !y) return KVAE_ERR_NULL;
  if (N < 1 || F < 1 || F > 128 || O < 1 || (int64_t)O * F > 12288 || (softmax && O > 16)) return KVAE_ERR_DIMS;
  kvae_rnn_launch_linear_fwd(x, x_stride, N, F, W, b, O, softmax, y, (hipStream_t)stream);
  return launch_status("k_linear_fwd");
}
int kvae_linear_bwd_input(const float *g, const float *y, int64_t N, int32_t F, const float *W, int32_t O, float *g_logit,
                          float *dx, int64_t dx_stride, void *stream) {
  if (!g || !W || !dx || (y && !g_logit)) return KVAE_ERR_NULL;
  if (N < 1 || F < 1 || F > 128 || O < 1 || (int64_t)O * F > 12288 || (y && O > 16)) return KVAE_ERR_DIMS;
  kvae_rnn_launch_linear_bwd_input(g, y, N, F, W, O, g_logit, dx, dx_stride, (hipStream_t)stream);
  return launch_status("k_linear_bwd_input");
}
}  // extern "C"

// ---------------------------------------------------------------------------------------------
// fused conv epilogues of the frame VAE (vae_epilogue.h)
// ---------------------------------------------------------------------------------------------
#include "vae_epilogue.h"

// out[c] = sum_r partials[r, c]: second stage of every deterministic two-stage reduction (bias / weight gradient partial
// rows).  HBM-bound (the 32->128 layers hand over 256 x 36864 floats = 38 MB): a lane owns four columns (dwordx4, 1 KiB per
// wave and row), the eight waves of a block take every eighth row, four rows in flight per wave, fold through LDS.
__device__ __forceinline__ void colsum_v4_body(const float *__restrict__ partials, float *__restrict__ out, int64_t rows, int64_t cols,
                                               unsigned block) {
  __shared__ float4 red[8][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t c = ((int64_t)block * 64 + lane) * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (c < cols) {
    const float *p = partials + c;
    int64_t r = wv;
    for (; r + 24 < rows; r += 32) {
      const float4 a = *reinterpret_cast<const float4 *>(p + r * cols), b = *reinterpret_cast<const float4 *>(p + (r + 8) * cols),
                   d = *reinterpret_cast<const float4 *>(p + (r + 16) * cols), e = *reinterpret_cast<const float4 *>(p + (r + 24) * cols);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += d.x; s2.y += d.y; s2.z += d.z; s2.w += d.w;
      s3.x += e.x; s3.y += e.y; s3.z += e.z; s3.w += e.w;
    }
    for (; r < rows; r += 8) {
      const float4 a = *reinterpret_cast<const float4 *>(p + r * cols);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
  }
  red[wv][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                              (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (wv == 0 && c < cols) {
    float4 t = red[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) { const float4 u = red[w][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4 *>(out + c) = t;
  }
}
__global__ __launch_bounds__(512) void k_colsum_v4(const float *__restrict__ partials, float *__restrict__ out, int64_t rows,
                                                   int64_t cols) {
  colsum_v4_body(partials, out, rows, cols, blockIdx.x);
}
// two jobs in one launch: the first nb_a workgroups take job a, the rest job b
__global__ __launch_bounds__(512) void k_colsum_v4_pair(const float *__restrict__ pa, float *__restrict__ oa, int64_t rows_a, int64_t cols_a,
                                                        unsigned nb_a, const float *__restrict__ pb, float *__restrict__ ob,
                                                        int64_t rows_b, int64_t cols_b) {
  if (blockIdx.x < nb_a) colsum_v4_body(pa, oa, rows_a, cols_a, blockIdx.x);
  else colsum_v4_body(pb, ob, rows_b, cols_b, blockIdx.x - nb_a);
}
// any column count (scalar): 64 columns x 4 row lanes per block
__global__ __launch_bounds__(256) void k_colsum(const float *__restrict__ partials, float *__restrict__ out, int64_t rows,
                                                int64_t cols) {
  __shared__ float red[256];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int64_t r = ry;
    for (; r + 4 < rows; r += 8) {
      s0 += partials[r * cols + c];
      s1 += partials[(r + 4) * cols + c];
    }
    if (r < rows) s0 += partials[r * cols + c];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (ry == 0 && c < cols) out[c] = (red[cx] + red[64 + cx]) + (red[128 + cx] + red[192 + cx]);
}

// Forward, R = 1 or 2: each thread produces 4 consecutive outputs along W (one 16-byte store) from one 16-byte
// (R = 1) or two 8-byte (R = 2: the two sub-pixel channels dx = 0,1 of this output row) loads; 32-bit index math.
template <int R>
__global__ __launch_bounds__(256) void k_vae_epilogue_fwd_v4(const float *__restrict__ in, const float *__restrict__ bias,
                                                             float *__restrict__ out, int C, int H, int W, int64_t quads,
                                                             int relu) {
  const int OW = W * R, OH = H * R, QW = OW / 4;
  for (int64_t qd = (int64_t)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += (int64_t)gridDim.x * 256) {
    const int qw = (int)(qd % QW);
    const int64_t t1 = qd / QW;
    const int oh = (int)(t1 % OH);
    const int64_t t2 = t1 / OH;
    const int c = (int)(t2 % C);
    const int64_t n = t2 / C;
    float4 v;
    if (R == 1) {
      const float b = bias[c];
      const float4 x = *reinterpret_cast<const float4 *>(in + ((n * C + c) * H + oh) * W + 4 * qw);
      v = make_float4(x.x + b, x.y + b, x.z + b, x.w + b);
    } else {
      const int ch0 = c * 4 + (oh & 1) * 2;
      const int64_t base = ((n * (C * 4) + ch0) * H + (oh >> 1)) * W + 2 * qw;
      const float2 a = *reinterpret_cast<const float2 *>(in + base);                    // dx = 0, w = 2qw, 2qw+1
      const float2 d = *reinterpret_cast<const float2 *>(in + base + (int64_t)H * W);  // dx = 1
      const float b0 = bias[ch0], b1 = bias[ch0 + 1];
      v = make_float4(a.x + b0, d.x + b1, a.y + b0, d.y + b1);
    }
    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    *reinterpret_cast<float4 *>(out + qd * 4) = v;
  }
}

__global__ __launch_bounds__(256) void k_vae_epilogue_fwd(const float *in, const float *bias, float *out, EpiShape s,
                                                          int64_t total, int relu) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256)
    epi_fwd_elem(s, in, bias, out, o, relu);
}

// Backward with the bias gradient folded in: a block owns 256 consecutive output positions of the per-sample volume
// [C, H*r, W*r] and walks a chunk of samples; per-thread sums go to per-channel LDS bins (ds_add_f32), one partial row
// per (chunk, channel) leaves the block -> bias_partials[chunk, C*r*r] (summed by the caller; fixed order per bin is not
// guaranteed inside a block: float LDS atomics, differences are at rounding level).
__global__ __launch_bounds__(256) void k_vae_epilogue_bwd_bias(const float *__restrict__ g_out, const float *__restrict__ out,
                                                               float *__restrict__ g_in, float *__restrict__ bias_partials,
                                                               EpiShape s, int n_per_chunk, int relu) {
  extern __shared__ float bins[];   // C*r*r floats
  const int Cin = s.C * s.r * s.r;
  const int OW = s.W * s.r, OH = s.H * s.r;
  const int64_t vol = (int64_t)s.C * OH * OW;
  for (int i = threadIdx.x; i < Cin; i += 256) bins[i] = 0.f;
  __syncthreads();
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;   // position inside one sample's output volume
  if (pos < vol) {
    const int ow = (int)(pos % OW);
    const int oh = (int)((pos / OW) % OH);
    const int c = (int)(pos / ((int64_t)OW * OH));
    const int ch = c * s.r * s.r + (oh % s.r) * s.r + (ow % s.r);
    const int64_t in_off = ((int64_t)ch * s.H + oh / s.r) * s.W + ow / s.r;
    const int64_t in_vol = (int64_t)Cin * s.H * s.W;
    const int64_t n0 = (int64_t)blockIdx.y * n_per_chunk;
    const int64_t n1 = n0 + n_per_chunk < s.N ? n0 + n_per_chunk : s.N;
    float acc = 0.f;
    for (int64_t n = n0; n < n1; ++n) {
      float g = g_out[n * vol + pos];
      if (relu && !(out[n * vol + pos] > 0.f)) g = 0.f;
      g_in[n * in_vol + in_off] = g;
      acc += g;
    }
    atomicAdd(&bins[ch], acc);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cin; i += 256) {
    // only the channels this block touched are non-zero; every block writes its own disjoint partial row slice
    if (bins[i] != 0.f) atomicAdd(&bias_partials[(int64_t)blockIdx.y * Cin + i], bins[i]);
  }
}

__global__ __launch_bounds__(256) void k_vae_epilogue_bwd(const float *g_out, const float *out, float *g_in, EpiShape s,
                                                          int64_t total, int relu) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256)
    epi_bwd_elem(s, g_out, out, g_in, o, relu);
}

#define KVAE_EPI_SAMPLES_PER_CHUNK 32
extern "C" int64_t kvae_bias_partial_rows(int64_t N);
static unsigned epi_grid(int64_t total) {
  const int64_t blocks = (total + 255) / 256;
  return (unsigned)(blocks < 256 * 32 ? blocks : 256 * 32);   // <= 32 blocks per CU, grid-stride the rest
}

extern "C" {
int kvae_bias_shuffle_act_fwd(const float *in, const float *bias, float *out, int64_t N, int32_t C, int32_t H, int32_t W,
                              int32_t r, int32_t relu, void *stream) {
  if (!in || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || r < 1) return KVAE_ERR_ARG;
  const EpiShape s{N, C, H, W, r};
  const int64_t total = N * C * H * W * r * r;
  const bool aligned = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (aligned && r == 1 && W % 4 == 0)
    k_vae_epilogue_fwd_v4<1><<<dim3(epi_grid(total / 4)), dim3(256), 0, (hipStream_t)stream>>>(in, bias, out, C, H, W, total / 4, relu);
  else if (aligned && r == 2 && W % 2 == 0)
    k_vae_epilogue_fwd_v4<2><<<dim3(epi_grid(total / 4)), dim3(256), 0, (hipStream_t)stream>>>(in, bias, out, C, H, W, total / 4, relu);
  else
    k_vae_epilogue_fwd<<<dim3(epi_grid(total)), dim3(256), 0, (hipStream_t)stream>>>(in, bias, out, s, total, relu);
  return launch_status("k_vae_epilogue_fwd");
}
int kvae_bias_shuffle_act_bwd(const float *g_out, const float *out, float *g_in, float *bias_partials, int64_t N, int32_t C,
                              int32_t H, int32_t W, int32_t r, int32_t relu, void *stream) {
  if (!g_out || !g_in || (relu && !out)) return KVAE_ERR_NULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || r < 1) return KVAE_ERR_ARG;
  const EpiShape s{N, C, H, W, r};
  const int64_t total = N * C * H * W * r * r;
  hipStream_t st = (hipStream_t)stream;
  if (bias_partials) {
    const int Cin = C * r * r;
    const int64_t chunks = kvae_bias_partial_rows(N);
    if (hipMemsetAsync(bias_partials, 0, sizeof(float) * chunks * Cin, st) != hipSuccess) return launch_status("memset bias_partials");
    const int64_t vol = (int64_t)C * H * W * r * r;
    k_vae_epilogue_bwd_bias<<<dim3((unsigned)((vol + 255) / 256), (unsigned)chunks), dim3(256), sizeof(float) * Cin, st>>>(
        g_out, out, g_in, bias_partials, s, KVAE_EPI_SAMPLES_PER_CHUNK, relu);
    return launch_status("k_vae_epilogue_bwd_bias");
  }
  k_vae_epilogue_bwd<<<dim3(epi_grid(total)), dim3(256), 0, st>>>(g_out, out, g_in, s, total, relu);
  return launch_status("k_vae_epilogue_bwd");
}
int kvae_colsum(const float *partials, float *out, int64_t rows, int64_t cols, void *stream) {
  if (!partials || !out) return KVAE_ERR_NULL;
  if (rows < 1 || cols < 1) return KVAE_ERR_ARG;
  if ((cols & 3) == 0 && ((((uintptr_t)partials | (uintptr_t)out) & 15) == 0))
    k_colsum_v4<<<dim3((unsigned)((cols + 255) / 256)), dim3(512), 0, (hipStream_t)stream>>>(partials, out, rows, cols);
  else
    k_colsum<<<dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, (hipStream_t)stream>>>(partials, out, rows, cols);
  return launch_status("k_colsum");
}
int kvae_colsum2(const float *pa, float *oa, int64_t rows_a, int64_t cols_a, const float *pb, float *ob, int64_t rows_b,
                 int64_t cols_b, void *stream) {
  if (!pa || !oa || !pb || !ob) return KVAE_ERR_NULL;
  if (rows_a < 1 || cols_a < 1 || rows_b < 1 || cols_b < 1) return KVAE_ERR_ARG;
  if (((cols_a | cols_b) & 3) == 0 && ((((uintptr_t)pa | (uintptr_t)oa | (uintptr_t)pb | (uintptr_t)ob) & 15) == 0)) {
    const unsigned nb_a = (unsigned)((cols_a + 255) / 256), nb_b = (unsigned)((cols_b + 255) / 256);
    k_colsum_v4_pair<<<dim3(nb_a + nb_b), dim3(512), 0, (hipStream_t)stream>>>(pa, oa, rows_a, cols_a, nb_a, pb, ob, rows_b, cols_b);
    return launch_status("k_colsum_v4_pair");
  }
  const int rc = kvae_colsum(pa, oa, rows_a, cols_a, stream);
  return rc ? rc : kvae_colsum(pb, ob, rows_b, cols_b, stream);
}
int64_t kvae_bias_partial_rows(int64_t N) { return (N + KVAE_EPI_SAMPLES_PER_CHUNK - 1) / KVAE_EPI_SAMPLES_PER_CHUNK; }
}  // extern "C"

// ---------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam on flat buffers: two launches instead of ~12 (norm, clamp, reciprocal, scale, three foreach kernels)
// ---------------------------------------------------------------------------------------------
constexpr int CA_BLOCKS = 512;   // partial sums of squares (ws[0..CA_BLOCKS))
constexpr int CA_MAX_SEG = 1024;  // parameter tensors ("segments") of one flat buffer
// Segments: seg_of[i] names the parameter tensor element i belongs to; a segment with seg_active[s] == 0 is a FROZEN parameter
// (requires_grad False: the reference's training phases, train.py:142-207) - torch's clip_grad_norm_ and Adam skip it because its
// .grad is None: it adds nothing to the norm, its moments and its step count stay as they are.  Every segment counts its own steps
// (torch keeps `step` per parameter, so a parameter thawed at epoch 6 starts its bias correction at step 1).
__global__ __launch_bounds__(256) void k_grad_sumsq(const float *__restrict__ g, int64_t n, const int32_t *__restrict__ seg_of,
                                                    int n_seg, const float *__restrict__ seg_active,
                                                    const float *__restrict__ div_dev, float *__restrict__ seg_steps,
                                                    float *__restrict__ ws) {
  __shared__ float red[256];
  const float inv = div_dev ? 1.0f / fmaxf(*div_dev, 1.0f) : 1.0f;
  const bool gated = seg_of && seg_active;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = g[i] * inv;
    if (gated && seg_active[seg_of[i]] == 0.f) v = 0.f;
    s = fmaf(v, v, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[blockIdx.x] = red[0];
  if (blockIdx.x == 0)                                  // the next launch reads the incremented step counts
    for (int sg = threadIdx.x; sg < n_seg; sg += 256)
      if (!seg_active || seg_active[sg] != 0.f) seg_steps[sg] += 1.0f;
}
__global__ __launch_bounds__(256) void k_clip_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, const int32_t *__restrict__ seg_of, int n_seg,
                                                   const float *__restrict__ seg_active, const float *__restrict__ seg_steps,
                                                   const float *__restrict__ lr_dev, float lr, float beta1, float beta2, float eps,
                                                   float wd, float clip, const float *__restrict__ div_dev,
                                                   float *__restrict__ norm_out, const float *__restrict__ ws, int nparts) {
  __shared__ float red[256];
  __shared__ float s_step_size[CA_MAX_SEG], s_bc2s[CA_MAX_SEG];   // step_size < 0 marks a frozen segment
  float s = 0.f;                                      // every block folds the same partials in the same order
  for (int i = threadIdx.x; i < nparts; i += 256) s += ws[i];
  red[threadIdx.x] = s;
  const float lrv = lr_dev ? *lr_dev : lr;
  for (int sg = threadIdx.x; sg < n_seg; sg += 256) {
    const float step = seg_steps[sg];
    const bool on = !seg_active || seg_active[sg] != 0.f;
    s_step_size[sg] = on ? lrv / (1.0f - powf(beta1, step)) : -1.0f;
    s_bc2s[sg] = on ? sqrtf(1.0f - powf(beta2, step)) : 1.0f;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float total = sqrtf(red[0]);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
  const float inv = div_dev ? 1.0f / fmaxf(*div_dev, 1.0f) : 1.0f;
  const float scale = inv * (clip > 0.f ? fminf(clip / (total + 1e-6f), 1.0f) : 1.0f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int sg = seg_of ? seg_of[i] : 0;
    const float step_size = s_step_size[sg], bc2s = s_bc2s[sg];
    if (step_size < 0.f) continue;
    float gi = g[i] * scale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);          // lerp
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi, v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) / bc2s + eps);
  }
}
extern "C" int kvae_clip_adam(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                              const int32_t *seg_of, int32_t n_seg, const float *seg_active, float *seg_steps, const float *lr_dev,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float clip, const float *div_dev,
                              float *norm_out, float *ws, void *stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !seg_steps || !ws) return KVAE_ERR_NULL;
  if (n < 1 || n_seg < 1 || n_seg > CA_MAX_SEG || (!seg_of && n_seg != 1)) return KVAE_ERR_ARG;
  const int64_t want = (n + 1023) / 1024;
  const unsigned parts = (unsigned)(want < CA_BLOCKS ? want : CA_BLOCKS);
  k_grad_sumsq<<<dim3(parts), dim3(256), 0, (hipStream_t)stream>>>(grads, n, seg_of, n_seg, seg_active, div_dev, seg_steps, ws);
  int rc = launch_status("k_grad_sumsq");
  if (rc) return rc;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  k_clip_adam<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, seg_of, n_seg, seg_active,
                                                                   seg_steps, lr_dev, lr, beta1, beta2, eps, weight_decay, clip,
                                                                   div_dev, norm_out, ws, (int)parts);
  return launch_status("k_clip_adam");
}

// ---------------------------------------------------------------------------------------------
// regime chain of the switching dynamics (regime.h)
// ---------------------------------------------------------------------------------------------
#include "regime.h"

__global__ __launch_bounds__(64) void k_regime_fwd(const float *logits, const float *init_logits, const float *gumbel,
                                                   const float *P, float *y_seq, float *log_q, float *log_p, int T, int K,
                                                   float tau, const float *tau_dev, int hard) {
  __shared__ RegimeLds L;
  if (tau_dev) tau = *tau_dev;   // device scalar: follows the schedule under hipGraph replay
  regime_fwd_body(logits, init_logits, gumbel, P, y_seq, log_q, log_p, blockIdx.x, T, K, tau, hard, L);
}
__global__ __launch_bounds__(64) void k_regime_bwd(const float *logits, const float *init_logits, const float *gumbel,
                                                   const float *P, const float *y_seq, const float *g_y, const float *g_lq,
                                                   const float *g_lp, float *g_logits, float *g_init, int T, int K, float tau,
                                                   const float *tau_dev) {
  __shared__ RegimeLds L;
  if (tau_dev) tau = *tau_dev;
  regime_bwd_body(logits, init_logits, gumbel, P, y_seq, g_y, g_lq, g_lp, g_logits, g_init, blockIdx.x, T, K, tau, L);
}

extern "C" {
int kvae_regime_fwd(const float *logits, const float *init_logits, const float *gumbel, const float *P, float *y_seq,
                    float *log_q, float *log_p, int32_t B, int32_t T, int32_t K, float tau, const float *tau_dev, int32_t hard,
                    void *stream) {
  if (!logits || !init_logits || !gumbel || !P || !y_seq || !log_q || !log_p) return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || K < 1 || K > KVAE_REGIME_MAX_K || (!tau_dev && !(tau > 0.f))) return KVAE_ERR_ARG;
  // 1 (default): lane-grid wavefront per sequence up to 4096 sequences of K <= 8, thread-per-sequence beyond; 2: thread-per-
  // sequence always; 0: the LDS wave-per-sequence bodies of regime.h (any K <= 16: also the fallback of the other two)
  static const int tpp_env = getenv("KVAE_REGIME_TPP") ? atoi(getenv("KVAE_REGIME_TPP")) : 1;
  if (tpp_env == 1 && kvae_grid_launch_regime_fwd(logits, init_logits, gumbel, P, y_seq, log_q, log_p, B, T, K, tau, tau_dev, hard,
                                                  (hipStream_t)stream))
    return launch_status("k_regime_fwd_grid");
  if (tpp_env && kvae_tpp_launch_regime_fwd(logits, init_logits, gumbel, P, y_seq, log_q, log_p, B, T, K, tau, tau_dev, hard,
                                            (hipStream_t)stream))
    return launch_status("k_regime_fwd_tpp");
  k_regime_fwd<<<dim3(B), dim3(64), 0, (hipStream_t)stream>>>(logits, init_logits, gumbel, P, y_seq, log_q, log_p, T, K, tau,
                                                              tau_dev, hard);
  return launch_status("k_regime_fwd");
}
int kvae_regime_bwd(const float *logits, const float *init_logits, const float *gumbel, const float *P, const float *y_seq,
                    const float *g_y, const float *g_log_q, const float *g_log_p, float *g_logits, float *g_init, int32_t B,
                    int32_t T, int32_t K, float tau, const float *tau_dev, void *stream) {
  if (!logits || !init_logits || !gumbel || !P || !y_seq || !g_y || !g_log_q || !g_log_p || !g_logits || !g_init)
    return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || K < 1 || K > KVAE_REGIME_MAX_K || (!tau_dev && !(tau > 0.f))) return KVAE_ERR_ARG;
  static const int tpp_env = getenv("KVAE_REGIME_TPP") ? atoi(getenv("KVAE_REGIME_TPP")) : 1;
  if (tpp_env == 1 && kvae_grid_launch_regime_bwd(logits, init_logits, gumbel, P, y_seq, g_y, g_log_q, g_log_p, g_logits, g_init, B, T, K,
                                                  tau, tau_dev, (hipStream_t)stream))
    return launch_status("k_regime_bwd_grid");
  if (tpp_env && kvae_tpp_launch_regime_bwd(logits, init_logits, gumbel, P, y_seq, g_y, g_log_q, g_log_p, g_logits, g_init, B, T, K, tau,
                                            tau_dev, (hipStream_t)stream))
    return launch_status("k_regime_bwd_tpp");
  k_regime_bwd<<<dim3(B), dim3(64), 0, (hipStream_t)stream>>>(logits, init_logits, gumbel, P, y_seq, g_y, g_log_q, g_log_p,
                                                             g_logits, g_init, T, K, tau, tau_dev);
  return launch_status("k_regime_bwd");
}
}  // extern "C"

// ---------------------------------------------------------------------------------------------
// bidirectional GRU of the regime posterior (gru_fast.h)
// ---------------------------------------------------------------------------------------------
#include "gru_fast.h"

extern "C" {
int kvae_bigru_fwd(const float *x, const float *const w_ih[2], const float *const w_hh[2], const float *const b_ih[2],
                   const float *const b_hh[2], float *h_seq, float *gates, int32_t B, int32_t T, int32_t I, int32_t H,
                   void *stream) {
  if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !h_seq || !gates) return KVAE_ERR_NULL;
  for (int d = 0; d < 2; ++d)
    if (!w_ih[d] || !w_hh[d] || !b_ih[d] || !b_hh[d]) return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || H != 50 || I != 2) return KVAE_ERR_DIMS;
  const GruWeights wf{w_ih[0], w_hh[0], b_ih[0], b_hh[0]}, wb{w_ih[1], w_hh[1], b_ih[1], b_hh[1]};
  k_gru_fwd_fast<50, 2><<<dim3(B, 2), dim3(256), 0, (hipStream_t)stream>>>(x, wf, wb, h_seq, gates, B, T);
  return launch_status("k_gru_fwd_fast");
}
int kvae_bigru_bwd(const float *g_h, const float *gates, const float *h_seq, const float *const w_ih[2],
                   const float *const w_hh[2], float *d_pre_i, float *d_pre_h, float *dx, int32_t B, int32_t T, int32_t I,
                   int32_t H, void *stream) {
  if (!g_h || !gates || !h_seq || !w_ih || !w_hh || !d_pre_i || !d_pre_h || !dx) return KVAE_ERR_NULL;
  for (int d = 0; d < 2; ++d)
    if (!w_ih[d] || !w_hh[d]) return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || H != 50 || I != 2) return KVAE_ERR_DIMS;
  const GruWeights wf{w_ih[0], w_hh[0], nullptr, nullptr}, wb{w_ih[1], w_hh[1], nullptr, nullptr};
  k_gru_bwd_fast<50, 2><<<dim3(B, 2), dim3(192), 0, (hipStream_t)stream>>>(g_h, gates, h_seq, wf, wb, d_pre_i, d_pre_h, dx, B, T);
  return launch_status("k_gru_bwd_fast");
}
}  // extern "C"

// ---------------------------------------------------------------------------------------------
// fused Bernoulli reconstruction term (vae_loss.h): one wavefront per frame
// ---------------------------------------------------------------------------------------------
#include "vae_loss.h"

__global__ __launch_bounds__(256) void k_vae_bce_fwd(const float *__restrict__ logits, const float *__restrict__ x,
                                                     float *__restrict__ frame_ll, int64_t frames, int pixels) {
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (f >= frames) return;
  const float *l = logits + f * pixels, *t = x + f * pixels;
  float acc = 0.f;
  if ((pixels & 3) == 0 && (((uintptr_t)l | (uintptr_t)t) & 15) == 0) {
    for (int i = lane * 4; i < pixels; i += 256) {
      const float4 a = *reinterpret_cast<const float4 *>(l + i), b = *reinterpret_cast<const float4 *>(t + i);
      acc += bce_logit(a.x, b.x) + bce_logit(a.y, b.y) + bce_logit(a.z, b.z) + bce_logit(a.w, b.w);
    }
  } else {
    for (int i = lane; i < pixels; i += 64) acc += bce_logit(l[i], t[i]);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) frame_ll[f] = -acc;
}

__global__ __launch_bounds__(256) void k_vae_bce_bwd(const float *__restrict__ logits, const float *__restrict__ x,
                                                     const float *__restrict__ g_frame, float *__restrict__ g_logits,
                                                     int64_t total, int pixels) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
    const float g = -g_frame[i / pixels];   // pixels % 4 == 0 on this path: the 4 elements share a frame
    const float4 a = *reinterpret_cast<const float4 *>(logits + i), b = *reinterpret_cast<const float4 *>(x + i);
    *reinterpret_cast<float4 *>(g_logits + i) = make_float4(g * (sigmoid_stable(a.x) - b.x), g * (sigmoid_stable(a.y) - b.y),
                                                            g * (sigmoid_stable(a.z) - b.z), g * (sigmoid_stable(a.w) - b.w));
  }
}
__global__ __launch_bounds__(256) void k_vae_bce_bwd_scalar(const float *logits, const float *x, const float *g_frame,
                                                            float *g_logits, int64_t total, int pixels) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    g_logits[i] = -g_frame[i / pixels] * (sigmoid_stable(logits[i]) - x[i]);
}

extern "C" {
int kvae_bce_frames_fwd(const float *logits, const float *x, float *frame_ll, int64_t frames, int32_t pixels, void *stream) {
  if (!logits || !x || !frame_ll) return KVAE_ERR_NULL;
  if (frames < 1 || pixels < 1) return KVAE_ERR_ARG;
  k_vae_bce_fwd<<<dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(logits, x, frame_ll, frames, pixels);
  return launch_status("k_vae_bce_fwd");
}
int kvae_bce_frames_bwd(const float *logits, const float *x, const float *g_frame, float *g_logits, int64_t frames,
                        int32_t pixels, void *stream) {
  if (!logits || !x || !g_frame || !g_logits) return KVAE_ERR_NULL;
  if (frames < 1 || pixels < 1) return KVAE_ERR_ARG;
  const int64_t total = frames * pixels;
  const bool v4 = (pixels & 3) == 0 && ((((uintptr_t)logits | (uintptr_t)x | (uintptr_t)g_logits) & 15) == 0);
  if (v4)
    k_vae_bce_bwd<<<dim3(epi_grid(total / 4)), dim3(256), 0, (hipStream_t)stream>>>(logits, x, g_frame, g_logits, total, pixels);
  else
    k_vae_bce_bwd_scalar<<<dim3(epi_grid(total)), dim3(256), 0, (hipStream_t)stream>>>(logits, x, g_frame, g_logits, total, pixels);
  return launch_status("k_vae_bce_bwd");
}
}  // extern "C"

#include "vae_conv_edge.h"
extern "C" {
int64_t kvae_conv_edge_partial_rows(int64_t N) { return N < 1024 ? (N < 1 ? 1 : N) : 1024; }   // four workgroups per CU

int kvae_dec_head_fwd(const float *in, const float *W, const float *bias, float *logits, float *w_scratch, int64_t N,
                      int32_t Cin, int32_t side, void *stream) {
  if (!in || !W || !bias || !logits || !w_scratch) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != DH_CI || side != DH_S) return KVAE_ERR_DIMS;
  k_dec_head_prep<<<dim3(1), dim3(256), 0, (hipStream_t)stream>>>(W, w_scratch);
  k_dec_head_fwd<<<dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream>>>(in, w_scratch, bias, logits);
  return launch_status("k_dec_head_fwd");
}
int kvae_dec_head_bwd(const float *in, const float *W, const float *g_logits, float *g_in, float *w_partials,
                      float *b_partials, float *w_scratch, int64_t N, int32_t Cin, int32_t side, void *stream) {
  if (!in || !W || !g_logits || !w_partials || !b_partials || !w_scratch) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != DH_CI || side != DH_S) return KVAE_ERR_DIMS;
  if (g_in) {
    k_dec_head_prep<<<dim3(1), dim3(256), 0, (hipStream_t)stream>>>(W, w_scratch);
    k_dec_head_bwd_data<<<dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream>>>(g_logits, w_scratch, g_in);
    const int rc = launch_status("k_dec_head_bwd_data");
    if (rc) return rc;
  }
  k_dec_head_wrw<<<dim3((unsigned)kvae_conv_edge_partial_rows(N)), dim3(256), 0, (hipStream_t)stream>>>(in, g_logits, w_partials,
                                                                                                     b_partials, N);
  return launch_status("k_dec_head_wrw");
}
int kvae_enc_stem_fwd(const float *x, const float *W, const float *bias, float *out, uint32_t *relu_bits, int64_t N, int32_t Cout,
                      int32_t side, void *stream) {
  if (!x || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cout != ES_CO || side != ES_IN) return KVAE_ERR_DIMS;
  k_enc_stem_fwd<<<dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream>>>(x, W, bias, out, relu_bits);
  return launch_status("k_enc_stem_fwd");
}
int kvae_enc_stem_bwd(const float *x, const float *out, const uint32_t *relu_bits, const float *g_out, float *w_partials,
                      float *b_partials, int64_t N, int32_t Cout, int32_t side, void *stream) {
  if (!x || (!out && !relu_bits) || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cout != ES_CO || side != ES_IN) return KVAE_ERR_DIMS;
  static const int mfma = getenv("KVAE_STEM_MFMA") ? atoi(getenv("KVAE_STEM_MFMA")) : 1;   // 0: VALU version, 2: mask from out (A/B runs)
  const dim3 grid((unsigned)kvae_conv_edge_partial_rows(N));
  if (mfma == 1 && relu_bits) k_enc_stem_wrw_mfma<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, relu_bits, g_out, w_partials, b_partials, N);
  else if (!out) return KVAE_ERR_NULL;
  else if (mfma) k_enc_stem_wrw_mfma<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, relu_bits, g_out, w_partials, b_partials, N);
  else k_enc_stem_wrw<<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
  return launch_status("k_enc_stem_wrw");
}
}  // extern "C"

#include "vae_conv_mid.h"
static inline int64_t enc_mid_grid(int64_t N, int32_t side) {
  const int64_t fpi = side == 16 ? 2 : 8, iters = (N + fpi - 1) / fpi;
  return iters < 256 ? (iters < 1 ? 1 : iters) : 256;   // one persistent workgroup per CU
}
extern "C" {
int64_t kvae_enc_mid_partial_rows(int64_t N, int32_t side) { return enc_mid_grid(N, side); }

int kvae_enc_mid_fwd(const float *in, const float *W, const float *bias, float *out, int64_t N, int32_t C, int32_t side,
                     void *stream) {
  if (!in || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * EM_C * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;   // 32-bit byte offsets
  if (C != EM_C || (side != 16 && side != 8)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)enc_mid_grid(N, side));
  if (side == 16) k_enc_mid_fwd<16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, W, bias, out, N);
  else k_enc_mid_fwd<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, W, bias, out, N);
  return launch_status("k_enc_mid_fwd");
}
int kvae_enc_mid_bwd(const float *in, const float *W, const float *out, const float *g_out, float *g_in,
                     float *w_partials, float *b_partials, int64_t N, int32_t C, int32_t side, void *stream) {
  if (!in || !W || !out || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * EM_C * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;
  if (C != EM_C || (side != 16 && side != 8)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)enc_mid_grid(N, side));
  if (g_in) {
    if (side == 16) k_enc_mid_bwd_data<16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_in, N);
    else k_enc_mid_bwd_data<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_in, N);
    const int rc = launch_status("k_enc_mid_bwd_data");
    if (rc) return rc;
  }
  if (side == 16) k_enc_mid_wrw<16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, out, g_out, w_partials, b_partials, N);
  else k_enc_mid_wrw<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, out, g_out, w_partials, b_partials, N);
  return launch_status("k_enc_mid_wrw");
}
}  // extern "C"

#include "vae_conv_up_wino.h"
static bool dec_up_wino() {
  static const int env = getenv("KVAE_WINO") ? atoi(getenv("KVAE_WINO")) : 1;   // 0: direct convolution (A/B runs)
  return env != 0;
}
// Persistent decoder-block workgroups: one per CU, or fewer (KVAE_UP_WGS) so that a second stream's kernels that do not fit
// beside them (anything above 32 registers per lane) find free CUs while they run.
static int g_dec_up_wgs = 0;   // 0: not set through kvae_dec_up_set_workgroups
static inline int64_t dec_up_cap() {
  static const int env = getenv("KVAE_UP_WGS") ? atoi(getenv("KVAE_UP_WGS")) : 0;   // the environment wins over the setter (A/B runs)
  const int v = env >= 1 && env <= 256 ? env : __atomic_load_n(&g_dec_up_wgs, __ATOMIC_RELAXED);
  return v >= 1 && v <= 256 ? v : 256;
}
static inline int64_t dec_up_grid(int64_t N, int32_t side) {
  const int64_t fpi = side == 8 ? 2 : 8, iters = (N + fpi - 1) / fpi, cap = dec_up_cap();
  return iters < cap ? (iters < 1 ? 1 : iters) : cap;
}
extern "C" {
int64_t kvae_dec_up_partial_rows(int64_t N, int32_t side) { return dec_up_grid(N, side); }
int32_t kvae_dec_up_set_workgroups(int32_t n) {
  const int32_t prev = (int32_t)dec_up_cap();
  __atomic_store_n(&g_dec_up_wgs, n >= 1 && n <= 256 ? n : 0, __ATOMIC_RELAXED);
  return prev;
}

int kvae_dec_up_fwd(const float *x, const float *W, const float *bias, float *out, int64_t N, int32_t Cin, int32_t side,
                    void *stream) {
  if (!x || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * UP_CO * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;   // 32-bit byte offsets
  if (Cin != UP_CI || (side != 8 && side != 4)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)dec_up_grid(N, side));
  if (dec_up_wino()) {   // pairs of workgroups (one per half of the output channels) walk the column sets together
    const int64_t sets = side == 8 ? N : (N + 3) / 4;
    const dim3 wgrid((unsigned)(sets < dec_up_cap() ? sets : dec_up_cap()));
    if (side == 8) k_dec_up_fwd_wino<8><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
    else k_dec_up_fwd_wino<4><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
    return launch_status("k_dec_up_fwd_wino");
  }
  if (side == 8) k_dec_up_fwd<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
  else k_dec_up_fwd<4><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
  return launch_status("k_dec_up_fwd");
}
int kvae_dec_up_bwd(const float *x, const float *W, const float *out, const float *g_out, float *g_x, float *w_partials,
                    float *b_partials, int64_t N, int32_t Cin, int32_t side, void *stream) {
  if (!x || !W || !out || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * UP_CO * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;
  if (Cin != UP_CI || (side != 8 && side != 4)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)dec_up_grid(N, side));
  if (g_x) {
    const int64_t sets = side == 8 ? N : (N + 3) / 4;
    const dim3 wgrid((unsigned)(sets < dec_up_cap() ? sets : dec_up_cap()));
    if (dec_up_wino() && side == 8) k_dec_up_bwd_data_wino<8><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    else if (dec_up_wino()) k_dec_up_bwd_data_wino<4><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    else if (side == 8) k_dec_up_bwd_data<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    else k_dec_up_bwd_data<4><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    const int rc = launch_status("k_dec_up_bwd_data");
    if (rc) return rc;
  }
  if (dec_up_wino()) {
    if (side == 8) k_dec_up_wrw_wino<8><<<grid, dim3(512), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
    else k_dec_up_wrw_wino<4><<<grid, dim3(512), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
    return launch_status("k_dec_up_wrw_wino");
  }
  if (side == 8) k_dec_up_wrw<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
  else k_dec_up_wrw<4><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
  return launch_status("k_dec_up_wrw");
}
}  // extern "C"

#include "vae_heads.h"
extern "C" {
int64_t kvae_head_partial_rows(void) { return HD_WAVES; }

int kvae_enc_head_fwd(const float *feat, const float *Wmu, const float *bmu, const float *Wvar, const float *bvar,
                      const float *eps, float *mu, float *var, float *a, int64_t N, int32_t F, int32_t A,
                      float noise_emission, void *stream) {
  if (!feat || !Wmu || !bmu || !Wvar || !bvar || !mu || !var || !a) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  const unsigned grid = (unsigned)(N < 4 * 512 ? (N + 3) / 4 : 512);
  k_enc_head_fwd<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(feat, Wmu, bmu, Wvar, bvar, eps, noise_emission, mu, var, a, N);
  return launch_status("k_enc_head_fwd");
}
int kvae_enc_head_bwd(const float *feat, const float *Wmu, const float *Wvar, const float *var, const float *eps,
                      const float *g_a, const float *g_mu, const float *g_var, float *g_feat, float *w_partials,
                      float *b_partials, int64_t N, int32_t F, int32_t A, float noise_emission, void *stream) {
  if (!feat || !Wmu || !Wvar || !var || !g_feat || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  k_enc_head_bwd<<<dim3(HD_WAVES / 4), dim3(256), 0, (hipStream_t)stream>>>(feat, Wmu, Wvar, var, eps, g_a, g_mu, g_var,
                                                                             noise_emission, g_feat, w_partials, b_partials, N);
  return launch_status("k_enc_head_bwd");
}
int kvae_dec_fc_fwd(const float *a, const float *W, const float *b, float *h, int64_t N, int32_t F, int32_t A, void *stream) {
  if (!a || !W || !b || !h) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  k_dec_fc_fwd<<<dim3(epi_grid(N * (HD_F / 4))), dim3(256), 0, (hipStream_t)stream>>>(a, W, b, h, N * (HD_F / 4));
  return launch_status("k_dec_fc_fwd");
}
int kvae_dec_fc_bwd(const float *g_h, const float *a, const float *W, float *g_a, float *w_partials, float *b_partials,
                    int64_t N, int32_t F, int32_t A, void *stream) {
  if (!g_h || !a || !W || !g_a || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  k_dec_fc_bwd<<<dim3(HD_WAVES / 4), dim3(256), 0, (hipStream_t)stream>>>(g_h, a, W, g_a, w_partials, b_partials, N);
  return launch_status("k_dec_fc_bwd");
}
int kvae_latent_reg_fwd(const float *a, const float *mu, const float *var, float *reg, int64_t N, int32_t A, void *stream) {
  if (!a || !mu || !var || !reg) return KVAE_ERR_NULL;
  if (N < 1 || A < 1) return KVAE_ERR_ARG;
  k_latent_reg_fwd<<<dim3(epi_grid(N)), dim3(256), 0, (hipStream_t)stream>>>(a, mu, var, reg, N, A);
  return launch_status("k_latent_reg_fwd");
}
int kvae_latent_reg_bwd(const float *a, const float *mu, const float *var, const float *g, float *g_a, float *g_mu,
                        float *g_var, int64_t N, int32_t A, void *stream) {
  if (!a || !mu || !var || !g || !g_a || !g_mu || !g_var) return KVAE_ERR_NULL;
  if (N < 1 || A < 1) return KVAE_ERR_ARG;
  k_latent_reg_bwd<<<dim3(epi_grid(N * A)), dim3(256), 0, (hipStream_t)stream>>>(a, mu, var, g, g_a, g_mu, g_var, N, A);
  return launch_status("k_latent_reg_bwd");
}
int kvae_loss_head_fwd(const float *lpx, const float *regf, const float *mask, const float *elbo_kf, const float *beta,
                       float scale_reconstruction, float vae_weight, float kf_weight, const float *weights_dev, float *out6,
                       float *coef3, int64_t n, void *stream) {
  if (!lpx || !regf || !elbo_kf || !beta || !out6 || !coef3) return KVAE_ERR_NULL;
  if (n < 1) return KVAE_ERR_ARG;
  k_loss_head_fwd<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(lpx, regf, mask, elbo_kf, beta, scale_reconstruction, vae_weight,
                                                                  kf_weight, weights_dev, out6, coef3, n);
  return launch_status("k_loss_head_fwd");
}
int kvae_loss_head_bwd(const float *g_loss, const float *coef3, const float *mask, float *g_lpx, float *g_regf,
                       float *g_elbo_kf, int64_t n, void *stream) {
  if (!g_loss || !coef3 || !g_lpx || !g_regf || !g_elbo_kf) return KVAE_ERR_NULL;
  if (n < 1) return KVAE_ERR_ARG;
  k_loss_head_bwd<<<dim3(epi_grid(n)), dim3(256), 0, (hipStream_t)stream>>>(g_loss, coef3, mask, g_lpx, g_regf,
                                                                            g_elbo_kf, n);
  return launch_status("k_loss_head_bwd");
}
}  // extern "C"
