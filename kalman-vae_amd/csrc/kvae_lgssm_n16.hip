// kvae_lgssm_n16.hip — the (n, m, p) = (16, 16, 2) LGSSM kernels on the f32 matrix cores (lgssm_n16.h): BASELINE
// configs[4] ("stress": z = u = 16, T = 200).  One 64-lane wavefront per sequence; grid = B.
#include <hip/hip_runtime.h>

#include "lgssm_n16.h"
#include "lgssm_n16_elbo.h"
#include "lgssm_m4.h"

using namespace kvae;

template <bool AUX>   // AUX: also save the gains K | S | J per step for the backward (states.aux)
__global__ __launch_bounds__(64) void k_smooth_fwd_n16(kvae_lgssm_problem P, kvae_lgssm_states S, int do_filter, int do_rts) {
  __shared__ n16::Lds L;
  n16::smooth_fwd_wave<AUX>(P, S, do_filter, do_rts, L);
}

extern "C" void kvae_n16_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                    hipStream_t s) {
  if (st->aux)
    k_smooth_fwd_n16<true><<<dim3(p->B), dim3(64), 0, s>>>(*p, *st, do_filter, do_rts);
  else
    k_smooth_fwd_n16<false><<<dim3(p->B), dim3(64), 0, s>>>(*p, *st, do_filter, do_rts);
}

// HAS_FP: upstream gradients of the filtered / predicted stacks present; HAS_GQ: the caller wants gQ
template <bool HAS_FP, bool HAS_GQ>
__global__ __launch_bounds__(64) void k_smooth_bwd_n16(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
                                                       kvae_lgssm_input_grads G, float *ws) {
  __shared__ n16::Lds L;
  n16::smooth_bwd_wave<HAS_FP, HAS_GQ>(P, S, U, G, ws, L);
}

extern "C" void kvae_n16_launch_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                                    const kvae_lgssm_input_grads *out, float *ws, int has_fp, hipStream_t s) {
  const dim3 grid(p->B), block(64);
  const bool gq = out->gQ.ptr != nullptr;
  if (has_fp && gq) k_smooth_bwd_n16<true, true><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
  else if (has_fp) k_smooth_bwd_n16<true, false><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
  else if (gq) k_smooth_bwd_n16<false, true><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
  else k_smooth_bwd_n16<false, false><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
}

// ---- ELBO terms (lgssm_n16_elbo.h): grid = B*T, one wavefront per (sequence, step) ------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (observed, not promised: a wrong guess is only slower), and step t reads
// A, B, Q and z of step t + 1 as well: give every XCD one contiguous range of (b, t) so that the neighbour's operands are in
// ITS L2 instead of being fetched over the fabric a second time (bijective for any grid size).
__global__ __launch_bounds__(64) void k_elbo_probe_n16(kvae_lgssm_problem P, const float *Sig_s, const float *mus, const float *eps,
                                                       float *zst, int32_t *levels) {
  n16::elbo_probe_wave(P, Sig_s, mus, eps, zst, levels);
}
template <bool GRADS, bool HAS_GQ>
__global__ __launch_bounds__(64) void k_elbo_n16(kvae_lgssm_problem P, const float *mus, const float *Sigs, const float *eps,
                                                 float *terms, const int32_t *levels, const float *zst, float *g_mus,
                                                 float *g_Sigs, kvae_lgssm_input_grads G) {
  __shared__ n16::ELds L;
  n16::elbo_wave<GRADS, HAS_GQ>(P, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, G, L);
}

__global__ __launch_bounds__(64) void k_elbo_probe4_n16(kvae_lgssm_problem P, const float *Sig_s, const float *mus, const float *eps,
                                                        float *zst, int32_t *levels) {
  n16::elbo_probe4_wave(P, Sig_s, mus, eps, zst, levels);
}
template <bool GRADS>
__global__ __launch_bounds__(64) void k_elbo4_n16(kvae_lgssm_problem P, const float *mus, const float *Sigs, const float *eps,
                                                  float *terms, const int32_t *levels, const float *zst, float *g_mus,
                                                  float *g_Sigs, kvae_lgssm_input_grads G) {
  __shared__ n16::ELds4 L;
  n16::elbo4_wave<GRADS>(P, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, G, L);
}
__global__ __launch_bounds__(64) void k_elbo_zfix_n16(kvae_lgssm_problem P, const float *Sig_s, const float *mus, const float *eps,
                                                      float *zst, const int32_t *levels) {
  n16::elbo_zfix_wave(P, Sig_s, mus, eps, zst, levels);
}
// a Q shared by the whole batch (lstm dynamics): the four-steps-per-wavefront layout, unless the caller wants g Q per step
static bool elbo_shared_q(const kvae_lgssm_problem *p) { return p->Q.sb == 0 && p->Q.st == 0; }

extern "C" void kvae_n16_launch_elbo_probe(const kvae_lgssm_problem *p, const float *Sig_s, const float *mus, const float *eps,
                                           float *zst, int32_t *levels, hipStream_t s) {
  if (elbo_shared_q(p)) {
    k_elbo_probe4_n16<<<dim3((unsigned)((int64_t)p->B * ((p->T + 3) / 4))), dim3(64), 0, s>>>(*p, Sig_s, mus, eps, zst, levels);
  } else {
    k_elbo_probe_n16<<<dim3((unsigned)((int64_t)p->B * p->T)), dim3(64), 0, s>>>(*p, Sig_s, mus, eps, zst, levels);
  }
  k_elbo_zfix_n16<<<dim3((unsigned)((int64_t)p->B * ((p->T + 3) / 4))), dim3(64), 0, s>>>(*p, Sig_s, mus, eps, zst, levels);
}
extern "C" void kvae_n16_launch_elbo(const kvae_lgssm_problem *p, const float *mus, const float *Sigs, const float *eps,
                                     float *terms, const int32_t *levels, const float *zst, float *g_mus, float *g_Sigs,
                                     const kvae_lgssm_input_grads *g, int have_g, hipStream_t s) {
  const dim3 grid((unsigned)((int64_t)p->B * p->T)), block(64);
  if (elbo_shared_q(p) && !(have_g && g->gQ.ptr)) {
    const dim3 grid4((unsigned)((int64_t)p->B * ((p->T + 3) / 4)));
    if (have_g) k_elbo4_n16<true><<<grid4, block, 0, s>>>(*p, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, *g);
    else k_elbo4_n16<false><<<grid4, block, 0, s>>>(*p, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, *g);
    return;
  }
  if (!have_g) k_elbo_n16<false, false><<<grid, block, 0, s>>>(*p, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, *g);
  else if (g->gQ.ptr) k_elbo_n16<true, true><<<grid, block, 0, s>>>(*p, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, *g);
  else k_elbo_n16<true, false><<<grid, block, 0, s>>>(*p, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, *g);
}

// ---- (n, m, p) = (4, 4, 2): sixteen sequences per wavefront, rows on the lanes of a quad, 4x4 products on the matrix cores
// (lgssm_m4.h over the layout of lgssm_q4.h); grid = ceil(B / 16); a ragged last wavefront recomputes (and re-stores, identically)
// the last sequence: no branch ----
template <bool AUX>
__global__ __launch_bounds__(64) void k_smooth_fwd_m4(kvae_lgssm_problem P, kvae_lgssm_states S, int do_filter, int do_rts) {
  m4::smooth_fwd_wave<AUX>(P, S, do_filter, do_rts);
}
extern "C" void kvae_q4_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                   hipStream_t s) {
  const dim3 grid((unsigned)((p->B + 15) / 16)), block(64);
  if (st->aux) k_smooth_fwd_m4<true><<<grid, block, 0, s>>>(*p, *st, do_filter, do_rts);
  else k_smooth_fwd_m4<false><<<grid, block, 0, s>>>(*p, *st, do_filter, do_rts);
}

template <bool HAS_FP, bool HAS_GQ>
__global__ __launch_bounds__(64) void k_smooth_bwd_m4(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
                                                      kvae_lgssm_input_grads G, float *ws) {
  m4::smooth_bwd_wave<HAS_FP, HAS_GQ>(P, S, U, G, ws);
}
extern "C" void kvae_q4_launch_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                                   const kvae_lgssm_input_grads *out, float *ws, int has_fp, hipStream_t s) {
  const dim3 grid((unsigned)((p->B + 15) / 16)), block(64);
  const bool gq = out->gQ.ptr != nullptr;
  if (has_fp && gq) k_smooth_bwd_m4<true, true><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
  else if (has_fp) k_smooth_bwd_m4<true, false><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
  else if (gq) k_smooth_bwd_m4<false, true><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
  else k_smooth_bwd_m4<false, false><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws);
}
