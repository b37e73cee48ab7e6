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
extern "C" int kvae_launch_status(const char *what) { return launch_status(what); }   // for kvae_vae.hip

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
