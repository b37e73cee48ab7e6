// kvae_lgssm_elbo16.hip — the ELBO kernels of (n, m, p) = (16, 16, 2) (lgssm_n16_elbo.h): wavefronts over (sequence, step) or
// (sequence, four steps), four matrices per wavefront with rows on lanes.  A unit of its own because it wants the default
// scheduling strategy (see kvae_lgssm_n16.hip).
#include <hip/hip_runtime.h>

#include "lgssm_n16_elbo.h"

using namespace kvae;

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

