// hostsim.cpp — TEST-ONLY host build of the kernel bodies in kalman-vae_amd/csrc/*.h.
//
// Compiles the very same body headers with KVAE_HOSTSIM (KV_PAR = serial loop, KV_SYNC = no-op) and
// exports the C ABI of include/kvae_lgssm.h over HOST pointers, one "wavefront" at a time.  It lets
// the CPU-only test tier (a) run every kernel's arithmetic against the oracle and the goldens and
// (b) run it under -fsanitize=address,undefined (GPU sanitizers are unavailable on the pool).
// It is never loaded by the product package (kalman-vae_amd/kvae/_native.py loads only the gfx950
// library and raises if that is missing).
#define KVAE_HOSTSIM 1
#include <stdlib.h>
#include <string.h>

#include <memory>

#include "../../kalman-vae_amd/csrc/lgssm_bwd.h"
#include "../../kalman-vae_amd/csrc/lgssm_elbo.h"
#include "../../kalman-vae_amd/csrc/lgssm_fwd.h"
#include "../../kalman-vae_amd/csrc/lgssm_n4.h"
#include "../../kalman-vae_amd/csrc/mix.h"

using namespace kvae;

static int check_problem(const kvae_lgssm_problem *p) {
  if (!p) return KVAE_ERR_NULL;
  if (p->B < 1 || p->T < 1 || p->n < 1 || p->m < 1 || p->p < 1 || p->n > KVAE_MAX_DIM || p->m > KVAE_MAX_DIM ||
      p->p > KVAE_MAX_DIM)
    return KVAE_ERR_DIMS;
  if (!p->A.ptr || !p->Bm.ptr || !p->C.ptr || !p->Q.ptr || !p->R || !p->mu0 || !p->Sigma0 || !p->Y || !p->U)
    return KVAE_ERR_NULL;
  return KVAE_OK;
}

#define KVAE_DISPATCH(P, CALL)                              \
  do {                                                      \
    if ((P).n == 4 && (P).m == 4 && (P).p == 2) {          \
      using D = SDims<4, 4, 2>;                             \
      CALL;                                                 \
    } else if ((P).n == 16 && (P).m == 16 && (P).p == 2) { \
      using D = SDims<16, 16, 2>;                           \
      CALL;                                                 \
    } else {                                                \
      using D = RDims;                                      \
      CALL;                                                 \
    }                                                       \
  } while (0)

template <class D>
static void run_fwd(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int do_filter, int do_rts) {
  auto L = std::make_unique<FwdLds<D>>();
  const D d(P.n, P.m, P.p);
  for (int b = 0; b < P.B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));  // poison: NaNs expose reads of unwritten scratch
    if (do_filter) filter_sweep(d, P, S, b, *L);
    if (do_rts) rts_sweep(d, P, S, b, *L);
  }
}

static void run_fwd_n4(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int do_filter, int do_rts) {
  using D = SDims<4, 4, 2>;
  auto L = std::make_unique<N4Lds<4>>();
  const D d(P.n, P.m, P.p);
  for (int b = 0; b < P.B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));
    if (do_filter) filter_sweep_n4(d, P, S, b, *L);
    if (do_rts) rts_sweep_n4(d, P, S, b, *L);
  }
}

static void run_bwd_n4(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                       const kvae_lgssm_input_grads &G, float *ws, int with_rts) {
  using D = SDims<4, 4, 2>;
  auto L = std::make_unique<N4BwdLds<4>>();
  const D d(P.n, P.m, P.p);
  for (int b = 0; b < P.B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));
    if (with_rts)
      rts_bwd_sweep_n4(d, P, S, U, G, ws, b, *L);
    else
      filter_bwd_seed(d, P, U, G, ws, b);
    filter_bwd_sweep_n4(d, P, S, G, ws, b, *L);
  }
}

template <class D>
static void run_bwd(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                    const kvae_lgssm_input_grads &G, float *ws, int with_rts) {
  auto L = std::make_unique<BwdLds<D>>();
  const D d(P.n, P.m, P.p);
  for (int b = 0; b < P.B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));
    if (with_rts)
      rts_bwd_sweep(d, P, S, U, G, ws, b, *L);
    else
      filter_bwd_seed(d, P, U, G, ws, b);
    filter_bwd_sweep(d, P, S, G, ws, b, *L);
  }
}

template <class D>
static void run_elbo(const kvae_lgssm_problem &P, const float *mus, const float *Sigs, const float *eps, float *terms,
                     int32_t *levels, float *ws, float *g_mus, float *g_Sigs, const kvae_lgssm_input_grads *g) {
  auto L = std::make_unique<ElboLds<D>>();
  const D d(P.n, P.m, P.p);
  levels[0] = levels[1] = 0;
  for (int b = 0; b < P.B; ++b)
    for (int t = 0; t < P.T; ++t) {
      memset(L.get(), 0xFF, sizeof(*L));
      elbo_probe_body(d, P, Sigs, mus, eps, ws, levels, b, t, *L);
    }
  for (int b = 0; b < P.B; ++b)
    for (int t = 0; t < P.T; ++t) {
      memset(L.get(), 0xFF, sizeof(*L));
      elbo_body(d, P, mus, Sigs, eps, terms, levels, ws, g_mus, g_Sigs, g, b, t, *L);
    }
}

// wave_emu_kernels.cpp: the product's wavefront-level kernels ((4,4,2) on lgssm_m4.h, (16,16,2) on lgssm_n16.h) on emulated
// wavefronts.  Off by default (the bodies above are the fast way to check arithmetic); kvae_hostsim_wave_emu(1) routes the
// smoother entry points there for the shapes and alignments the GPU dispatch (kvae_lgssm.hip: q4_ok / n16_ok) sends to them.
extern "C" {
void kvae_wemu_fwd_n4(const kvae_lgssm_problem *, const kvae_lgssm_states *, int, int);
void kvae_wemu_bwd_n4(const kvae_lgssm_problem *, const kvae_lgssm_states *, const kvae_lgssm_states *, const kvae_lgssm_input_grads *,
                      float *, int);
void kvae_wemu_fwd_n16(const kvae_lgssm_problem *, const kvae_lgssm_states *, int, int);
void kvae_wemu_bwd_n16(const kvae_lgssm_problem *, const kvae_lgssm_states *, const kvae_lgssm_states *,
                       const kvae_lgssm_input_grads *, float *, int);
}
static int g_wave_emu = 0;
static bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool stack16(const kvae_stack &s) { return al16(s.ptr) && s.sb % 4 == 0 && s.st % 4 == 0; }
static bool gstack16(const kvae_gstack &g) { return !g.ptr || (al16(g.ptr) && g.sb % 4 == 0 && g.st % 4 == 0); }
// the gates of the GPU dispatch (kvae_lgssm.hip: q4_ok / n16_ok and the alignment tests of launch_fwd / kvae_lgssm_smooth_bwd)
static bool wave_fwd_ok(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int n) {
  if (!g_wave_emu || p->n != n || p->m != n || p->p != 2) return false;
  const bool prob = stack16(p->A) && stack16(p->Bm) && stack16(p->Q) && stack16(p->C) && al16(p->Sigma0) && p->Sigma0_sb % 4 == 0 &&
                    al16(p->mu0) && p->mu0_sb % 4 == 0 && al16(p->U) && (reinterpret_cast<uintptr_t>(p->Y) & 7) == 0;
  return prob && al16(st->mus_filt) && al16(st->Sigmas_filt) && al16(st->mus_pred) && al16(st->Sigmas_pred) &&
         al16(st->mus_smooth) && al16(st->Sigmas_smooth) && al16(st->aux);
}
static bool wave_bwd_ok(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                        const kvae_lgssm_input_grads *out, const float *ws, int n) {
  return wave_fwd_ok(p, saved, n) && saved->aux && al16(ws) && al16(up->Sigmas_smooth) && al16(up->Sigmas_filt) &&
         al16(up->Sigmas_pred) && gstack16(out->gA) && gstack16(out->gB) && gstack16(out->gQ) && al16(out->g_Sigma0) && out->gU;
}

extern "C" {

int kvae_hostsim_wave_emu(int on) {
  const int was = g_wave_emu;
  g_wave_emu = on;
  return was;
}

static int fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *st, int do_filter, int do_rts) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!st || !st->mus_filt || !st->Sigmas_filt || !st->mus_pred || !st->Sigmas_pred) return KVAE_ERR_NULL;
  if (do_rts && (!st->mus_smooth || !st->Sigmas_smooth)) return KVAE_ERR_NULL;
  if (wave_fwd_ok(prob, st, 4)) {
    kvae_wemu_fwd_n4(prob, st, do_filter, do_rts);
    return KVAE_OK;
  }
  if (wave_fwd_ok(prob, st, 16)) {
    kvae_wemu_fwd_n16(prob, st, do_filter, do_rts);
    return KVAE_OK;
  }
  if (prob->n == 4 && prob->m == 4 && prob->p == 2 && (st->aux || !do_filter)) {
    run_fwd_n4(*prob, *st, do_filter, do_rts);
    return KVAE_OK;
  }
  KVAE_DISPATCH(*prob, (run_fwd<D>(*prob, *st, do_filter, do_rts)));
  return KVAE_OK;
}
int kvae_lgssm_filter_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, void *) { return fwd(prob, out, 1, 0); }
int kvae_lgssm_rts_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *io, void *) { return fwd(prob, io, 0, 1); }
int kvae_lgssm_smooth_fwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *out, void *) { return fwd(prob, out, 1, 1); }

int kvae_lgssm_smooth_bwd(const kvae_lgssm_problem *prob, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                          const kvae_lgssm_input_grads *out, float *ws, int with_rts, void *) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!saved || !up || !out || !ws) return KVAE_ERR_NULL;
  if (!out->gA.ptr || !out->gB.ptr || !out->gC.ptr || !out->gY) return KVAE_ERR_NULL;
  if (with_rts && (wave_bwd_ok(prob, saved, up, out, ws, 4) || wave_bwd_ok(prob, saved, up, out, ws, 16))) {
    // as kvae_lgssm.hip: upstream gradients of the filtered / predicted stacks come all four or not at all
    const int fp = (up->mus_filt != nullptr) + (up->Sigmas_filt != nullptr) + (up->mus_pred != nullptr) + (up->Sigmas_pred != nullptr);
    if ((fp == 0 || fp == 4) && up->mus_smooth && up->Sigmas_smooth) {
      if (prob->n == 4) kvae_wemu_bwd_n4(prob, saved, up, out, ws, fp == 4);
      else kvae_wemu_bwd_n16(prob, saved, up, out, ws, fp == 4);
      return KVAE_OK;
    }
  }
  if (prob->n == 4 && prob->m == 4 && prob->p == 2 && saved->aux) {
    run_bwd_n4(*prob, *saved, *up, *out, ws, with_rts);
    return KVAE_OK;
  }
  KVAE_DISPATCH(*prob, (run_bwd<D>(*prob, *saved, *up, *out, ws, with_rts)));
  return KVAE_OK;
}

int kvae_lgssm_elbo(const kvae_lgssm_problem *prob, const float *mus_smooth, const float *Sigmas_smooth, const float *eps,
                    float *terms, int32_t *chol_levels, float *ws_lz, float *g_mus, float *g_Sigmas,
                    const kvae_lgssm_input_grads *g, void *) {
  int rc = check_problem(prob);
  if (rc) return rc;
  if (!mus_smooth || !Sigmas_smooth || !eps || !terms || !chol_levels) return KVAE_ERR_NULL;
  if (g_mus && (!g_Sigmas || !g || !g->gA.ptr || !g->gB.ptr || !g->gC.ptr || !g->gY)) return KVAE_ERR_NULL;
  chol_levels[2] = 0;   // kernel family: the generic bodies
  KVAE_DISPATCH(*prob, (run_elbo<D>(*prob, mus_smooth, Sigmas_smooth, eps, terms, chol_levels, ws_lz, g_mus, g_Sigmas, g)));
  return KVAE_OK;
}

int kvae_mix_fwd(const float *alpha, const float *base, float *out, int64_t rows, int32_t K, int32_t E, void *) {
  if (!alpha || !base || !out) return KVAE_ERR_NULL;
  if (rows < 1 || K < 1 || K > KVAE_MAX_K || E < 1) return KVAE_ERR_ARG;
  for (int64_t i = 0; i < rows * E; ++i) mix_fwd_elem(alpha, base, out, i, K, E);
  return KVAE_OK;
}
int64_t kvae_mix_bwd_partials(int64_t rows) { return (rows + KVAE_MIX_ROWS_PER_BLOCK - 1) / KVAE_MIX_ROWS_PER_BLOCK; }
int kvae_mix_bwd(const float *alpha, const float *base, const float *g_out, float *g_alpha, float *g_base, float *partials,
                 int64_t rows, int32_t K, int32_t E, int32_t accumulate_alpha, void *) {
  if (!alpha || !base || !g_out || !g_alpha || !g_base || !partials) return KVAE_ERR_NULL;
  if (rows < 1 || K < 1 || K > KVAE_MAX_K || E < 1) return KVAE_ERR_ARG;
  for (int64_t i = 0; i < rows * K; ++i) mix_bwd_alpha_elem(base, g_out, g_alpha, i, K, E, accumulate_alpha);
  const int64_t nblk = kvae_mix_bwd_partials(rows);
  for (int64_t blk = 0; blk < nblk; ++blk)
    for (int e = 0; e < E; ++e) mix_bwd_partial_elem(alpha, g_out, partials, blk, e, rows, K, E);
  for (int i = 0; i < K * E; ++i) mix_bwd_final_elem(partials, g_base, i, nblk, K * E);
  return KVAE_OK;
}
int kvae_abi_version(void) { return KVAE_ABI_VERSION; }
const char *kvae_last_error(void) { return ""; }
const char *kvae_build_info(void) { return "kvae_lgssm HOSTSIM (test-only host build of the kernel bodies)"; }
}

#include "../../kalman-vae_amd/csrc/lstm.h"
extern "C" {
int kvae_lstm_fwd(const float *x, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh, float *h_seq,
                  float *gates, float *c_seq, int32_t B, int32_t T, int32_t I, int32_t H, void *) {
  if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !h_seq || !gates || !c_seq) return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || I < 1 || I > KVAE_LSTM_MAX_I || H < 1 || H > KVAE_LSTM_MAX_H) return KVAE_ERR_DIMS;
  auto L = std::make_unique<LstmLds>();
  for (int b = 0; b < B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));
    lstm_fwd_body(x, w_ih, w_hh, b_ih, b_hh, h_seq, gates, c_seq, b, T, I, H, *L);
  }
  return KVAE_OK;
}
int kvae_lstm_bwd(const float *g_h, const float *gates, const float *c_seq, const float *w_ih, const float *w_hh,
                  float *d_pre, float *dx, int32_t B, int32_t T, int32_t I, int32_t H, void *) {
  if (!g_h || !gates || !c_seq || !w_ih || !w_hh || !d_pre || !dx) return KVAE_ERR_NULL;
  if (B < 1 || T < 1 || I < 1 || I > KVAE_LSTM_MAX_I || H < 1 || H > KVAE_LSTM_MAX_H) return KVAE_ERR_DIMS;
  auto L = std::make_unique<LstmLds>();
  for (int b = 0; b < B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));
    lstm_bwd_body(g_h, gates, c_seq, w_ih, w_hh, d_pre, dx, b, T, I, H, *L);
  }
  return KVAE_OK;
}
}

#include "../../kalman-vae_amd/csrc/vae_epilogue.h"
extern "C" {
int kvae_bias_shuffle_act_fwd(const float *in, const float *bias, float *out, int64_t N, int32_t C, int32_t H, int32_t W,
                              int32_t r, int32_t relu, void *) {
  if (!in || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || r < 1) return KVAE_ERR_ARG;
  const EpiShape s{N, C, H, W, r};
  for (int64_t o = 0; o < N * C * H * W * r * r; ++o) epi_fwd_elem(s, in, bias, out, o, relu);
  return KVAE_OK;
}
int64_t kvae_bias_partial_rows(int64_t N) { return (N + 31) / 32; }
int kvae_colsum(const float *partials, float *out, int64_t rows, int64_t cols, void *) {
  if (!partials || !out) return KVAE_ERR_NULL;
  if (rows < 1 || cols < 1) return KVAE_ERR_ARG;
  for (int64_t c = 0; c < cols; ++c) {
    float s = 0.f;
    for (int64_t r = 0; r < rows; ++r) s += partials[r * cols + c];
    out[c] = s;
  }
  return KVAE_OK;
}
int kvae_clip_adam(float *p, const float *g, float *m, float *v, int64_t n, const int32_t *seg_of, int32_t n_seg,
                   const float *seg_active, float *seg_steps, const float *lr_dev, float lr, float beta1, float beta2, float eps,
                   float wd, float clip, const float *div_dev, float *norm_out, float *ws, void *) {
  if (!p || !g || !m || !v || !seg_steps || !ws) return KVAE_ERR_NULL;
  if (n < 1 || n_seg < 1 || n_seg > 1024 || (!seg_of && n_seg != 1)) return KVAE_ERR_ARG;
  const float inv = div_dev ? 1.0f / std::fmax(*div_dev, 1.0f) : 1.0f;
  auto on = [&](int64_t i) { return !seg_active || seg_active[seg_of ? seg_of[i] : 0] != 0.f; };
  double ss = 0.0;
  for (int64_t i = 0; i < n; ++i)
    if (on(i)) ss += (double)(g[i] * inv) * (g[i] * inv);
  const float total = (float)std::sqrt(ss);
  if (norm_out) *norm_out = total;
  for (int s = 0; s < n_seg; ++s)
    if (!seg_active || seg_active[s] != 0.f) seg_steps[s] += 1.0f;
  const float scale = inv * (clip > 0.f ? std::fmin(clip / (total + 1e-6f), 1.0f) : 1.0f), lrv = lr_dev ? *lr_dev : lr;
  for (int64_t i = 0; i < n; ++i) {
    if (!on(i)) continue;
    const float step = seg_steps[seg_of ? seg_of[i] : 0];
    const float bc1 = 1.0f - std::pow(beta1, step), bc2s = std::sqrt(1.0f - std::pow(beta2, step)), step_size = lrv / bc1;
    float gi = g[i] * scale;
    if (wd != 0.f) gi += wd * p[i];
    m[i] = m[i] + (gi - m[i]) * (1.0f - beta1);
    v[i] = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    p[i] -= step_size * m[i] / (std::sqrt(v[i]) / bc2s + eps);
  }
  return KVAE_OK;
}
int kvae_colsum2(const float *pa, float *oa, int64_t rows_a, int64_t cols_a, const float *pb, float *ob, int64_t rows_b,
                 int64_t cols_b, void *s) {
  const int rc = kvae_colsum(pa, oa, rows_a, cols_a, s);
  return rc ? rc : kvae_colsum(pb, ob, rows_b, cols_b, s);
}
int kvae_bias_shuffle_act_bwd(const float *g_out, const float *out, float *g_in, float *bias_partials, int64_t N, int32_t C,
                              int32_t H, int32_t W, int32_t r, int32_t relu, void *) {
  if (!g_out || !g_in || (relu && !out)) return KVAE_ERR_NULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || r < 1) return KVAE_ERR_ARG;
  const EpiShape s{N, C, H, W, r};
  const int64_t vol = (int64_t)C * H * W * r * r;
  for (int64_t o = 0; o < N * vol; ++o) epi_bwd_elem(s, g_out, out, g_in, o, relu);
  if (bias_partials) {
    const int Cin = C * r * r;
    memset(bias_partials, 0, sizeof(float) * kvae_bias_partial_rows(N) * Cin);
    for (int64_t n = 0; n < N; ++n)
      for (int ch = 0; ch < Cin; ++ch)
        for (int k = 0; k < H * W; ++k) bias_partials[(n / 32) * Cin + ch] += g_in[(n * Cin + ch) * H * W + k];
  }
  return KVAE_OK;
}
}

#include "../../kalman-vae_amd/csrc/regime.h"
extern "C" {
int kvae_regime_fwd(const float *logits, const float *init_logits, const float *gumbel, const float *P, float *y_seq,
                    float *log_q, float *log_p, int32_t B, int32_t T, int32_t K, float tau, const float *tau_dev, int32_t hard,
                    void *) {
  if (!logits || !init_logits || !gumbel || !P || !y_seq || !log_q || !log_p) return KVAE_ERR_NULL;
  if (tau_dev) tau = *tau_dev;
  if (B < 1 || T < 1 || K < 1 || K > KVAE_REGIME_MAX_K || !(tau > 0.f)) return KVAE_ERR_ARG;
  auto L = std::make_unique<RegimeLds>();
  for (int b = 0; b < B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));
    regime_fwd_body(logits, init_logits, gumbel, P, y_seq, log_q, log_p, b, T, K, tau, hard, *L);
  }
  return KVAE_OK;
}
int kvae_regime_bwd(const float *logits, const float *init_logits, const float *gumbel, const float *P, const float *y_seq,
                    const float *g_y, const float *g_log_q, const float *g_log_p, float *g_logits, float *g_init, int32_t B,
                    int32_t T, int32_t K, float tau, const float *tau_dev, void *) {
  if (!logits || !init_logits || !gumbel || !P || !y_seq || !g_y || !g_log_q || !g_log_p || !g_logits || !g_init)
    return KVAE_ERR_NULL;
  if (tau_dev) tau = *tau_dev;
  if (B < 1 || T < 1 || K < 1 || K > KVAE_REGIME_MAX_K || !(tau > 0.f)) return KVAE_ERR_ARG;
  auto L = std::make_unique<RegimeLds>();
  for (int b = 0; b < B; ++b) {
    memset(L.get(), 0xFF, sizeof(*L));
    regime_bwd_body(logits, init_logits, gumbel, P, y_seq, g_y, g_log_q, g_log_p, g_logits, g_init, b, T, K, tau, *L);
  }
  return KVAE_OK;
}
}

// The bidirectional GRU has only the register-resident gfx950 kernels (gru_fast.h); the host simulation reports the
// shape as unsupported so that the Python side keeps nn.GRU on host tensors.
extern "C" {
int kvae_bigru_fwd(const float *, const float *const *, const float *const *, const float *const *, const float *const *,
                   float *, float *, int32_t, int32_t, int32_t, int32_t, void *) { return KVAE_ERR_DIMS; }
int kvae_bigru_bwd(const float *, const float *, const float *, const float *const *, const float *const *, float *, float *,
                   float *, int32_t, int32_t, int32_t, int32_t, void *) { return KVAE_ERR_DIMS; }
}

#include "../../kalman-vae_amd/csrc/vae_loss.h"
extern "C" {
int kvae_bce_frames_fwd(const float *logits, const float *x, float *frame_ll, int64_t frames, int32_t pixels, void *) {
  if (!logits || !x || !frame_ll) return KVAE_ERR_NULL;
  if (frames < 1 || pixels < 1) return KVAE_ERR_ARG;
  for (int64_t f = 0; f < frames; ++f) {
    float acc = 0.f;
    for (int i = 0; i < pixels; ++i) acc += bce_logit(logits[f * pixels + i], x[f * pixels + i]);
    frame_ll[f] = -acc;
  }
  return KVAE_OK;
}
int kvae_bce_frames_bwd(const float *logits, const float *x, const float *g_frame, float *g_logits, int64_t frames,
                        int32_t pixels, void *) {
  if (!logits || !x || !g_frame || !g_logits) return KVAE_ERR_NULL;
  if (frames < 1 || pixels < 1) return KVAE_ERR_ARG;
  for (int64_t i = 0; i < frames * pixels; ++i) g_logits[i] = -g_frame[i / pixels] * (sigmoid_stable(logits[i]) - x[i]);
  return KVAE_OK;
}
}

// In-kernel alpha-network filter: gfx950-only (register-resident LSTM rows); the simulator reports it unsupported so
// that the Python side keeps the per-step path on host tensors.
extern "C" int kvae_lgssm_filter_alpha_lstm(const kvae_lgssm_problem *, const kvae_lgssm_states *, const float *, const float *,
                                            const float *, const float *, const float *, const float *, const float *,
                                            const float *, const float *, int32_t, int32_t, float *, float *, float *, float *,
                                            float *, float *, void *) {
  return KVAE_ERR_DIMS;
}
extern "C" int kvae_lgssm_alpha_lstm_bwd(const kvae_lgssm_problem *, const kvae_lgssm_states *, const kvae_lgssm_states *,
                                         const kvae_lgssm_input_grads *, float *, int, const float *, const float *, const float *,
                                         const float *, const float *, const float *, int32_t, int32_t, const float *, const float *,
                                         const float *, const float *, const float *, float *, float *, float *, void *) {
  return KVAE_ERR_DIMS;
}

// Direct convolutions of the VAE's thin layers: plain loops with the same argument checks as the HIP launchers.
extern "C" {
int64_t kvae_conv_edge_partial_rows(int64_t N) { return N < 1024 ? (N < 1 ? 1 : N) : 1024; }

int kvae_dec_head_fwd(const float *in, const float *W, const float *bias, float *logits, float *w_scratch, int64_t N,
                      int32_t Cin, int32_t side, void *) {
  if (!in || !W || !bias || !logits || !w_scratch) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != 32 || side != 16) return KVAE_ERR_DIMS;
  const int S = side;
  for (int64_t n = 0; n < N; ++n)
    for (int co = 0; co < 4; ++co)
      for (int h = 0; h < S; ++h)
        for (int w = 0; w < S; ++w) {
          float acc = bias[co];
          for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int y = h + ky - 1, x = w + kx - 1;
                if (y < 0 || y >= S || x < 0 || x >= S) continue;
                acc += W[((co * Cin + ci) * 3 + ky) * 3 + kx] * in[((n * Cin + ci) * S + y) * S + x];
              }
          logits[n * 4 * S * S + (2 * h + co / 2) * 2 * S + 2 * w + (co & 1)] = acc;
        }
  return KVAE_OK;
}
int kvae_dec_head_bwd(const float *in, const float *W, const float *g_logits, float *g_in, float *w_partials,
                      float *b_partials, float *w_scratch, int64_t N, int32_t Cin, int32_t side, void *) {
  if (!in || !W || !g_logits || !w_partials || !b_partials || !w_scratch) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != 32 || side != 16) return KVAE_ERR_DIMS;
  const int S = side;
  const int64_t rows = kvae_conv_edge_partial_rows(N);
  memset(w_partials, 0, sizeof(float) * rows * 4 * Cin * 9);
  memset(b_partials, 0, sizeof(float) * rows * 4);
  if (g_in) memset(g_in, 0, sizeof(float) * N * Cin * S * S);
  for (int64_t n = 0; n < N; ++n) {
    float *wp = w_partials + (n % rows) * 4 * Cin * 9, *bp = b_partials + (n % rows) * 4;
    for (int co = 0; co < 4; ++co)
      for (int h = 0; h < S; ++h)
        for (int w = 0; w < S; ++w) {
          const float g = g_logits[n * 4 * S * S + (2 * h + co / 2) * 2 * S + 2 * w + (co & 1)];
          bp[co] += g;
          for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int y = h + ky - 1, x = w + kx - 1;
                if (y < 0 || y >= S || x < 0 || x >= S) continue;
                const int64_t ii = ((n * Cin + ci) * S + y) * S + x;
                const int wi = ((co * Cin + ci) * 3 + ky) * 3 + kx;
                wp[wi] += g * in[ii];
                if (g_in) g_in[ii] += g * W[wi];
              }
        }
  }
  return KVAE_OK;
}
int kvae_enc_stem_fwd(const float *x, const float *W, const float *bias, float *out, uint32_t *, int64_t N, int32_t Cout,
                      int32_t side, void *) {
  if (!x || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cout != 32 || side != 32) return KVAE_ERR_DIMS;
  const int S = side, O = side / 2;
  for (int64_t n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int oh = 0; oh < O; ++oh)
        for (int ow = 0; ow < O; ++ow) {
          float acc = bias[co];
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int y = 2 * oh + ky - 1, xx = 2 * ow + kx - 1;
              if (y < 0 || y >= S || xx < 0 || xx >= S) continue;
              acc += W[co * 9 + ky * 3 + kx] * x[(n * S + y) * S + xx];
            }
          out[((n * Cout + co) * O + oh) * O + ow] = acc > 0.f ? acc : 0.f;
        }
  return KVAE_OK;
}
int kvae_enc_stem_bwd(const float *x, const float *out, const uint32_t *, const float *g_out, float *w_partials,
                      float *b_partials, int64_t N, int32_t Cout, int32_t side, void *) {   // (mask always from out here)
  if (!x || !out || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cout != 32 || side != 32) return KVAE_ERR_DIMS;
  const int S = side, O = side / 2;
  const int64_t rows = kvae_conv_edge_partial_rows(N);
  memset(w_partials, 0, sizeof(float) * rows * Cout * 9);
  memset(b_partials, 0, sizeof(float) * rows * Cout);
  for (int64_t n = 0; n < N; ++n) {
    float *wp = w_partials + (n % rows) * Cout * 9, *bp = b_partials + (n % rows) * Cout;
    for (int co = 0; co < Cout; ++co)
      for (int oh = 0; oh < O; ++oh)
        for (int ow = 0; ow < O; ++ow) {
          const int64_t oi = ((n * Cout + co) * O + oh) * O + ow;
          if (!(out[oi] > 0.f)) continue;
          const float g = g_out[oi];
          bp[co] += g;
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const int y = 2 * oh + ky - 1, xx = 2 * ow + kx - 1;
              if (y < 0 || y >= S || xx < 0 || xx >= S) continue;
              wp[co * 9 + ky * 3 + kx] += g * x[(n * S + y) * S + xx];
            }
        }
  }
  return KVAE_OK;
}
}

// Encoder middle layers (stride-2 3x3, 32 -> 32, + ReLU): plain loops with the HIP launchers' argument checks.
extern "C" {
static int64_t enc_mid_grid_sim(int64_t N, int32_t side) {
  const int64_t fpi = side == 16 ? 2 : 8, iters = (N + fpi - 1) / fpi;
  return iters < 256 ? (iters < 1 ? 1 : iters) : 256;
}
int64_t kvae_enc_mid_partial_rows(int64_t N, int32_t side) { return enc_mid_grid_sim(N, side); }

int kvae_enc_mid_fwd(const float *in, const float *W, const float *bias, float *out, int64_t N, int32_t C, int32_t side,
                     void *) {
  if (!in || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (C != 32 || (side != 16 && side != 8)) return KVAE_ERR_DIMS;
  const int S = side, O = side / 2;
  for (int64_t n = 0; n < N; ++n)
    for (int co = 0; co < C; ++co)
      for (int oh = 0; oh < O; ++oh)
        for (int ow = 0; ow < O; ++ow) {
          float acc = bias[co];
          for (int ci = 0; ci < C; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int y = 2 * oh + ky - 1, x = 2 * ow + kx - 1;
                if (y < 0 || y >= S || x < 0 || x >= S) continue;
                acc += W[((co * C + ci) * 3 + ky) * 3 + kx] * in[((n * C + ci) * S + y) * S + x];
              }
          out[((n * C + co) * O + oh) * O + ow] = acc > 0.f ? acc : 0.f;
        }
  return KVAE_OK;
}
int kvae_enc_mid_bwd(const float *in, const float *W, const float *out, const float *g_out, float *g_in,
                     float *w_partials, float *b_partials, int64_t N, int32_t C, int32_t side, void *) {
  if (!in || !W || !out || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (C != 32 || (side != 16 && side != 8)) return KVAE_ERR_DIMS;
  const int S = side, O = side / 2;
  const int64_t rows = kvae_enc_mid_partial_rows(N, side);
  memset(w_partials, 0, sizeof(float) * rows * C * C * 9);
  memset(b_partials, 0, sizeof(float) * rows * C);
  if (g_in) memset(g_in, 0, sizeof(float) * N * C * S * S);
  for (int64_t n = 0; n < N; ++n) {
    float *wp = w_partials + (n % rows) * C * C * 9, *bp = b_partials + (n % rows) * C;
    for (int co = 0; co < C; ++co)
      for (int oh = 0; oh < O; ++oh)
        for (int ow = 0; ow < O; ++ow) {
          const int64_t oi = ((n * C + co) * O + oh) * O + ow;
          if (!(out[oi] > 0.f)) continue;
          const float g = g_out[oi];
          bp[co] += g;
          for (int ci = 0; ci < C; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int y = 2 * oh + ky - 1, x = 2 * ow + kx - 1;
                if (y < 0 || y >= S || x < 0 || x >= S) continue;
                const int64_t ii = ((n * C + ci) * S + y) * S + x;
                const int wi = ((co * C + ci) * 3 + ky) * 3 + kx;
                wp[wi] += g * in[ii];
                if (g_in) g_in[ii] += g * W[wi];
              }
        }
  }
  return KVAE_OK;
}
}

// Decoder up-sampling blocks (conv 32 -> 128, 3x3 pad 1, PixelShuffle(2), ReLU): plain loops, same argument checks.
extern "C" {
static int g_dec_up_wgs_sim = 256;
static int64_t dec_up_grid_sim(int64_t N, int32_t side) {
  const int64_t fpi = side == 8 ? 2 : 8, iters = (N + fpi - 1) / fpi;
  return iters < g_dec_up_wgs_sim ? (iters < 1 ? 1 : iters) : g_dec_up_wgs_sim;
}
int64_t kvae_dec_up_partial_rows(int64_t N, int32_t side) { return dec_up_grid_sim(N, side); }
int32_t kvae_dec_up_set_workgroups(int32_t n) {
  const int32_t prev = g_dec_up_wgs_sim;
  g_dec_up_wgs_sim = n >= 1 && n <= 256 ? n : 256;
  return prev;
}

int kvae_dec_up_fwd(const float *x, const float *W, const float *bias, float *out, int64_t N, int32_t Cin, int32_t side,
                    void *) {
  if (!x || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != 32 || (side != 8 && side != 4)) return KVAE_ERR_DIMS;
  const int S = side, O = 2 * side;
  for (int64_t n = 0; n < N; ++n)
    for (int co = 0; co < 128; ++co)
      for (int h = 0; h < S; ++h)
        for (int w = 0; w < S; ++w) {
          float acc = bias[co];
          for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int y = h + ky - 1, xx = w + kx - 1;
                if (y < 0 || y >= S || xx < 0 || xx >= S) continue;
                acc += W[((co * Cin + ci) * 3 + ky) * 3 + kx] * x[((n * Cin + ci) * S + y) * S + xx];
              }
          const int c = co / 4, dy = (co % 4) / 2, dx = co % 2;
          out[((n * 32 + c) * O + 2 * h + dy) * O + 2 * w + dx] = acc > 0.f ? acc : 0.f;
        }
  return KVAE_OK;
}
int kvae_dec_up_bwd(const float *x, const float *W, const float *out, const float *g_out, float *g_x, float *w_partials,
                    float *b_partials, int64_t N, int32_t Cin, int32_t side, void *) {
  if (!x || !W || !out || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != 32 || (side != 8 && side != 4)) return KVAE_ERR_DIMS;
  const int S = side, O = 2 * side;
  const int64_t rows = kvae_dec_up_partial_rows(N, side);
  memset(w_partials, 0, sizeof(float) * rows * 128 * Cin * 9);
  memset(b_partials, 0, sizeof(float) * rows * 128);
  if (g_x) memset(g_x, 0, sizeof(float) * N * Cin * S * S);
  for (int64_t n = 0; n < N; ++n) {
    float *wp = w_partials + (n % rows) * 128 * Cin * 9, *bp = b_partials + (n % rows) * 128;
    for (int co = 0; co < 128; ++co)
      for (int h = 0; h < S; ++h)
        for (int w = 0; w < S; ++w) {
          const int c = co / 4, dy = (co % 4) / 2, dx = co % 2;
          const int64_t oi = ((n * 32 + c) * O + 2 * h + dy) * O + 2 * w + dx;
          if (!(out[oi] > 0.f)) continue;
          const float g = g_out[oi];
          bp[co] += g;
          for (int ci = 0; ci < Cin; ++ci)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int y = h + ky - 1, xx = w + kx - 1;
                if (y < 0 || y >= S || xx < 0 || xx >= S) continue;
                const int64_t ii = ((n * Cin + ci) * S + y) * S + xx;
                const int wi = ((co * Cin + ci) * 3 + ky) * 3 + kx;
                wp[wi] += g * x[ii];
                if (g_x) g_x[ii] += g * W[wi];
              }
        }
  }
  return KVAE_OK;
}
}

// Skinny fc ends of the VAE and the latent regulariser: plain loops, same argument checks as the HIP launchers.
#include <math.h>
extern "C" {
int64_t kvae_head_partial_rows(void) { return 512; }

int kvae_enc_head_fwd(const float *feat, const float *Wmu, const float *bmu, const float *Wvar, const float *bvar,
                      const float *eps, float *mu, float *var, float *a, int64_t N, int32_t F, int32_t A, float ne, void *) {
  if (!feat || !Wmu || !bmu || !Wvar || !bvar || !mu || !var || !a) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != 512 || A != 2) return KVAE_ERR_DIMS;
  for (int64_t n = 0; n < N; ++n)
    for (int j = 0; j < A; ++j) {
      float dm = bmu[j], dv = bvar[j];
      for (int k = 0; k < F; ++k) { dm += feat[n * F + k] * Wmu[j * F + k]; dv += feat[n * F + k] * Wvar[j * F + k]; }
      const float s = ne * (1.f / (1.f + expf(-dv)));
      mu[n * A + j] = dm;
      var[n * A + j] = s;
      a[n * A + j] = eps ? dm + eps[n * A + j] * sqrtf(s + 1e-6f) : dm;
    }
  return KVAE_OK;
}
int kvae_enc_head_bwd(const float *feat, const float *Wmu, const float *Wvar, const float *var, const float *eps,
                      const float *g_a, const float *g_mu, const float *g_var, float *g_feat, float *w_partials,
                      float *b_partials, int64_t N, int32_t F, int32_t A, float ne, void *) {
  if (!feat || !Wmu || !Wvar || !var || !g_feat || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != 512 || A != 2) return KVAE_ERR_DIMS;
  const int64_t rows = kvae_head_partial_rows();
  memset(w_partials, 0, sizeof(float) * rows * 2 * A * F);
  memset(b_partials, 0, sizeof(float) * rows * 2 * A);
  for (int64_t n = 0; n < N; ++n) {
    float *wp = w_partials + (n % rows) * 2 * A * F, *bp = b_partials + (n % rows) * 2 * A;
    for (int k = 0; k < F; ++k) g_feat[n * F + k] = 0.f;
    for (int j = 0; j < A; ++j) {
      const float ga = g_a ? g_a[n * A + j] : 0.f, s = var[n * A + j];
      const float gm = (g_mu ? g_mu[n * A + j] : 0.f) + ga;
      const float gv = (g_var ? g_var[n * A + j] : 0.f) + (eps ? ga * eps[n * A + j] * 0.5f / sqrtf(s + 1e-6f) : 0.f);
      const float gs = gv * s * (1.f - s / ne);
      for (int k = 0; k < F; ++k) {
        g_feat[n * F + k] += gm * Wmu[j * F + k] + gs * Wvar[j * F + k];
        wp[j * F + k] += gm * feat[n * F + k];
        wp[(A + j) * F + k] += gs * feat[n * F + k];
      }
      bp[j] += gm;
      bp[A + j] += gs;
    }
  }
  return KVAE_OK;
}
int kvae_dec_fc_fwd(const float *a, const float *W, const float *b, float *h, int64_t N, int32_t F, int32_t A, void *) {
  if (!a || !W || !b || !h) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != 512 || A != 2) return KVAE_ERR_DIMS;
  for (int64_t n = 0; n < N; ++n)
    for (int k = 0; k < F; ++k) h[n * F + k] = a[n * A] * W[k * A] + a[n * A + 1] * W[k * A + 1] + b[k];
  return KVAE_OK;
}
int kvae_dec_fc_bwd(const float *g_h, const float *a, const float *W, float *g_a, float *w_partials, float *b_partials,
                    int64_t N, int32_t F, int32_t A, void *) {
  if (!g_h || !a || !W || !g_a || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != 512 || A != 2) return KVAE_ERR_DIMS;
  const int64_t rows = kvae_head_partial_rows();
  memset(w_partials, 0, sizeof(float) * rows * F * A);
  memset(b_partials, 0, sizeof(float) * rows * F);
  for (int64_t n = 0; n < N; ++n) {
    float *wp = w_partials + (n % rows) * F * A, *bp = b_partials + (n % rows) * F;
    float d0 = 0.f, d1 = 0.f;
    for (int k = 0; k < F; ++k) {
      const float g = g_h[n * F + k];
      d0 += g * W[k * A];
      d1 += g * W[k * A + 1];
      wp[k * A] += g * a[n * A];
      wp[k * A + 1] += g * a[n * A + 1];
      bp[k] += g;
    }
    g_a[n * A] = d0;
    g_a[n * A + 1] = d1;
  }
  return KVAE_OK;
}
int kvae_latent_reg_fwd(const float *a, const float *mu, const float *var, float *reg, int64_t N, int32_t A, void *) {
  if (!a || !mu || !var || !reg) return KVAE_ERR_NULL;
  if (N < 1 || A < 1) return KVAE_ERR_ARG;
  for (int64_t n = 0; n < N; ++n) {
    float s = 0.f;
    for (int j = 0; j < A; ++j) {
      const float x = a[n * A + j], d = x - mu[n * A + j], v = var[n * A + j];
      s += (-0.5f * x * x) - (-0.5f * logf(v) - d * d / (2.f * v));
    }
    reg[n] = s;
  }
  return KVAE_OK;
}
int kvae_latent_reg_bwd(const float *a, const float *mu, const float *var, const float *g, float *g_a, float *g_mu,
                        float *g_var, int64_t N, int32_t A, void *) {
  if (!a || !mu || !var || !g || !g_a || !g_mu || !g_var) return KVAE_ERR_NULL;
  if (N < 1 || A < 1) return KVAE_ERR_ARG;
  for (int64_t i = 0; i < N * A; ++i) {
    const float x = a[i], d = x - mu[i], v = var[i], gg = g[i / A];
    g_a[i] = gg * (-x + d / v);
    g_mu[i] = gg * (-d / v);
    g_var[i] = gg * (0.5f / v - d * d / (2.f * v * v));
  }
  return KVAE_OK;
}
}

// Scalar head of the objective: plain loops.
extern "C" {
int kvae_loss_head_fwd(const float *lpx, const float *regf, const float *mask, const float *elbo_kf, const float *beta,
                       float scale, float vae_w, float kf_w, const float *w_dev, float *out, float *coef, int64_t n, void *) {
  if (!lpx || !regf || !elbo_kf || !beta || !out || !coef) return KVAE_ERR_NULL;
  if (n < 1) return KVAE_ERR_ARG;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int64_t i = 0; i < n; ++i) {
    const float mk = mask ? mask[i] : 1.f;
    s0 += lpx[i] * mk; s1 += regf[i] * mk; s2 += mk;
  }
  if (w_dev) vae_w = w_dev[0], kf_w = w_dev[1];
  const float denom = s2 > 1.f ? s2 : 1.f, recon = s0 / denom, reg = s1 / denom;
  const float vae = scale * recon + beta[0] * reg, tot = vae_w * vae + kf_w * elbo_kf[0];
  out[0] = -tot; out[1] = tot; out[2] = elbo_kf[0]; out[3] = vae; out[4] = recon; out[5] = reg;
  coef[0] = -vae_w * scale / denom;
  coef[1] = -vae_w * beta[0] / denom;
  coef[2] = kf_w;
  return KVAE_OK;
}
int kvae_loss_head_bwd(const float *g, const float *coef, const float *mask, float *g_lpx, float *g_regf, float *g_kf,
                       int64_t n, void *) {
  if (!g || !coef || !g_lpx || !g_regf || !g_kf) return KVAE_ERR_NULL;
  if (n < 1) return KVAE_ERR_ARG;
  for (int64_t i = 0; i < n; ++i) {
    const float mk = mask ? mask[i] : 1.f;
    g_lpx[i] = g[0] * coef[0] * mk;
    g_regf[i] = g[0] * coef[1] * mk;
  }
  g_kf[0] = -coef[2] * g[0];
  return KVAE_OK;
}

int kvae_lgssm_emission_means(const kvae_lgssm_problem *prob, const float *ms, const float *mf, float *a_s, float *a_f, void *) {
  if (!prob || !prob->C.ptr || (!ms != !a_s) || (!mf != !a_f) || (!a_s && !a_f)) return KVAE_ERR_NULL;
  if (prob->B < 1 || prob->T < 1 || prob->n < 1 || prob->p < 1 || prob->n > KVAE_MAX_DIM || prob->p > KVAE_MAX_DIM) return KVAE_ERR_DIMS;
  const int n = prob->n, p = prob->p;
  for (int64_t b = 0; b < prob->B; ++b)
    for (int64_t t = 0; t < prob->T; ++t)
      for (int i = 0; i < p; ++i) {
        const float *c = prob->C.ptr + b * prob->C.sb + t * prob->C.st + (int64_t)i * n;
        const int64_t q = b * prob->T + t;
        float s = 0.f, f = 0.f;
        for (int k = 0; k < n; ++k) {
          if (ms) s = std::fma(c[k], ms[q * n + k], s);
          if (mf) f = std::fma(c[k], mf[q * n + k], f);
        }
        if (a_s) a_s[q * p + i] = s;
        if (a_f) a_f[q * p + i] = f;
      }
  return KVAE_OK;
}

// ---- alpha-network parameter gradients and linear heads: plain-loop twins of rnn_wgrad.h / small_linear.h --------------
int64_t kvae_rnn_wgrad_ws_floats(const kvae_wgrad_problem *, int32_t) { return 1; }
int kvae_rnn_wgrad(const kvae_wgrad_problem *probs, int32_t n, float *ws, void *) {
  if (!probs || !ws) return KVAE_ERR_NULL;
  if (n < 1 || n > 4) return KVAE_ERR_ARG;
  for (int i = 0; i < n; ++i) {
    const kvae_wgrad_problem &p = probs[i];
    const int C = p.H + p.I + (p.bias ? 1 : 0);
    if (!p.d || (p.H > 0 && !p.h) || (p.I > 0 && !p.x)) return KVAE_ERR_NULL;
    if (p.N < 1 || p.R < 1 || p.R > 256 || p.H < 0 || p.I < 0 || C < 1 || C > 256 || p.T < 1 || p.shift < -1 || p.shift > 1 ||
        p.N % p.T != 0)
      return KVAE_ERR_ARG;
    for (int r = 0; r < p.R; ++r)
      for (int c = 0; c < C; ++c) {
        double s = 0.0;
        for (int64_t q = 0; q < p.N; ++q) {
          float xv;
          if (c < p.H) {
            const int t = (int)(q % p.T) + p.shift;
            xv = (t >= 0 && t < p.T) ? p.h[(q + p.shift) * p.h_stride + c] : 0.f;
          } else if (c < p.H + p.I) {
            xv = p.x[q * p.x_stride + (c - p.H)];
          } else {
            xv = 1.f;
          }
          s += (double)p.d[q * p.d_stride + r] * xv;
        }
        if (c < p.H) { if (p.g_wh) p.g_wh[(int64_t)r * p.H + c] = (float)s; }
        else if (c < p.H + p.I) { if (p.g_wx) p.g_wx[(int64_t)r * p.I + (c - p.H)] = (float)s; }
        else if (p.g_b) p.g_b[r] = (float)s;
      }
  }
  return KVAE_OK;
}
int kvae_linear_fwd(const float *x, int64_t xs, int64_t N, int32_t F, const float *W, const float *b, int32_t O, int32_t softmax,
                    float *y, void *) {
  if (!x || !W || !y) return KVAE_ERR_NULL;
  if (N < 1 || F < 1 || F > 128 || O < 1 || (int64_t)O * F > 12288 || (softmax && O > 16)) return KVAE_ERR_DIMS;
  for (int64_t n = 0; n < N; ++n) {
    float mx = -INFINITY;
    for (int o = 0; o < O; ++o) {
      float acc = b ? b[o] : 0.f;
      for (int f = 0; f < F; ++f) acc = std::fma(x[n * xs + f], W[o * F + f], acc);
      y[n * O + o] = acc;
      mx = std::fmax(mx, acc);
    }
    if (softmax) {
      float sum = 0.f;
      for (int o = 0; o < O; ++o) sum += (y[n * O + o] = std::exp(y[n * O + o] - mx));
      for (int o = 0; o < O; ++o) y[n * O + o] /= sum;
    }
  }
  return KVAE_OK;
}
int kvae_linear_bwd_input(const float *g, const float *y, int64_t N, int32_t F, const float *W, int32_t O, float *g_logit, float *dx,
                          int64_t dxs, void *) {
  if (!g || !W || !dx || (y && !g_logit)) return KVAE_ERR_NULL;
  if (N < 1 || F < 1 || F > 128 || O < 1 || (int64_t)O * F > 12288 || (y && O > 16)) return KVAE_ERR_DIMS;
  for (int64_t n = 0; n < N; ++n) {
    float gl[256], dot = 0.f;
    for (int o = 0; o < O && y; ++o) dot = std::fma(g[n * O + o], y[n * O + o], dot);
    for (int o = 0; o < O; ++o) {
      gl[o] = y ? y[n * O + o] * (g[n * O + o] - dot) : g[n * O + o];
      if (y) g_logit[n * O + o] = gl[o];
    }
    for (int f = 0; f < F; ++f) {
      float acc = 0.f;
      for (int o = 0; o < O; ++o) acc = std::fma(gl[o], W[o * F + f], acc);
      dx[n * dxs + f] = acc;
    }
  }
  return KVAE_OK;
}
}
