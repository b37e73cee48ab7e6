// kvae_lgssm_n16.hip — the T-deep recursions that run one wavefront per SIMD: the (n, m, p) = (16, 16, 2) filter / smoother /
// adjoint on the f32 matrix cores (lgssm_n16.h: BASELINE configs[4], "stress": z = u = 16, T = 200; one 64-lane wavefront per
// sequence, grid = B) and the (4, 4, 2) ones on the quad layout (lgssm_m4.h: sixteen sequences per wavefront).  Their time is the
// dependent instruction stream of a single wavefront, so this unit is compiled with -mllvm -amdgpu-sched-strategy=max-ilp
// (__graft_entry__.UNIT_FLAGS); the n = 16 ELBO kernels, which are bound by VALU issue at full occupancy and
// lose 6 % under that strategy, live in kvae_lgssm_elbo16.hip.
#include <hip/hip_runtime.h>

#include "lgssm_n16.h"
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

// ---- (n, m, p) = (4, 4, 2): sixteen sequences per wavefront, rows on the lanes of a quad, 4x4 products on the matrix cores
// (lgssm_m4.h over the layout of lgssm_q4.h); grid = ceil(B / 16); a ragged last wavefront recomputes (and re-stores, identically)
// the last sequence: no branch ----
template <bool AUX>
__global__ __launch_bounds__(64) void k_smooth_fwd_m4(kvae_lgssm_problem P, kvae_lgssm_states S, int do_filter, int do_rts) {
  m4::smooth_fwd_wave<AUX>(P, S, do_filter, do_rts);
}
template <bool AUX>
__global__ __launch_bounds__(64) void k_gains_m4(kvae_lgssm_problem P, kvae_lgssm_states S) {
  m4::gains_wave<AUX>(P, S);
}
template <bool AUX>
static void launch_fwd_m4(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts, hipStream_t s) {
  const dim3 grid((unsigned)((p->B + 15) / 16)), block(64);
  if (m4::kv_m4_split(*p, do_filter, do_rts)) {   // filter sweep | all gains at once | smoother sweep (lgssm_m4.h: gains_wave)
    k_smooth_fwd_m4<AUX><<<grid, block, 0, s>>>(*p, *st, 1, 0);
    k_gains_m4<AUX><<<dim3(m4::kv_m4_gain_grid(*p)), block, 0, s>>>(*p, *st);
    k_smooth_fwd_m4<AUX><<<grid, block, 0, s>>>(*p, *st, 0, KV_M4_RTS_WITH_GAINS);
    return;
  }
  k_smooth_fwd_m4<AUX><<<grid, block, 0, s>>>(*p, *st, do_filter, do_rts);
}
extern "C" void kvae_q4_launch_fwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *st, int do_filter, int do_rts,
                                   hipStream_t s) {
  if (st->aux) launch_fwd_m4<true>(p, st, do_filter, do_rts, s);
  else launch_fwd_m4<false>(p, st, do_filter, do_rts, s);
}

template <bool HAS_FP, bool HAS_GQ>
__global__ __launch_bounds__(64) void k_smooth_bwd_m4(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
                                                      kvae_lgssm_input_grads G, float *ws, int part) {
  m4::smooth_bwd_wave<HAS_FP, HAS_GQ>(P, S, U, G, ws, part);
}
template <bool HAS_FP>
__global__ __launch_bounds__(64) void k_rts_bwd_items_m4(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_states U,
                                                         kvae_lgssm_input_grads G, float *ws) {
  m4::rts_bwd_items<HAS_FP>(P, S, U, G, ws);
}
__global__ __launch_bounds__(64) void k_filter_bwd_items_m4(kvae_lgssm_problem P, kvae_lgssm_states S, kvae_lgssm_input_grads G,
                                                           const float *ws) {
  m4::filter_bwd_items(P, S, G, ws);
}
template <bool HAS_FP, bool HAS_GQ>
static void launch_bwd_m4(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                          const kvae_lgssm_input_grads *out, float *ws, hipStream_t s) {
  const dim3 grid((unsigned)((p->B + 15) / 16)), block(64);
  if (m4::kv_m4_split_bwd(*p)) {   // each adjoint's chain alone, then what hangs off it for all steps at once (lgssm_m4.h)
    k_smooth_bwd_m4<HAS_FP, HAS_GQ><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws, KV_M4_BWD_CHAIN);
    k_rts_bwd_items_m4<HAS_FP><<<dim3(m4::kv_m4_gain_grid(*p)), block, 0, s>>>(*p, *saved, *up, *out, ws);
    k_smooth_bwd_m4<HAS_FP, HAS_GQ><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws, KV_M4_BWD_FCHAIN);
    k_filter_bwd_items_m4<<<dim3(m4::kv_m4_item_grid(*p)), block, 0, s>>>(*p, *saved, *out, ws);
    return;
  }
  k_smooth_bwd_m4<HAS_FP, HAS_GQ><<<grid, block, 0, s>>>(*p, *saved, *up, *out, ws, KV_M4_BWD_ALL);
}
extern "C" void kvae_q4_launch_bwd(const kvae_lgssm_problem *p, const kvae_lgssm_states *saved, const kvae_lgssm_states *up,
                                   const kvae_lgssm_input_grads *out, float *ws, int has_fp, hipStream_t s) {
  const bool gq = out->gQ.ptr != nullptr;
  if (has_fp && gq) launch_bwd_m4<true, true>(p, saved, up, out, ws, s);
  else if (has_fp) launch_bwd_m4<true, false>(p, saved, up, out, ws, s);
  else if (gq) launch_bwd_m4<false, true>(p, saved, up, out, ws, s);
  else launch_bwd_m4<false, false>(p, saved, up, out, ws, s);
}
