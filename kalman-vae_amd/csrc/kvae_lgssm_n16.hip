// kvae_lgssm_n16.hip — the (n, m, p) = (16, 16, 2) LGSSM kernels on the f32 matrix cores (lgssm_n16.h): BASELINE
// configs[4] ("stress": z = u = 16, T = 200).  One 64-lane wavefront per sequence; grid = B.
#include <hip/hip_runtime.h>

#include "lgssm_n16.h"

using namespace kvae;

template <bool AUX>   // AUX: also save the gains K | S | J per step for the backward (states.aux)
__global__ __launch_bounds__(64) void k_smooth_fwd_n16(kvae_lgssm_problem P, kvae_lgssm_states S, int do_filter, int do_rts) {
  __shared__ n16::Lds L;
  const int b = blockIdx.x;
  if (do_filter) {
    n16::filter_sweep<AUX>(P, S, b, L);
    __syncthreads();   // the smoother reads back what this wavefront has just written
  }
  if (do_rts) n16::rts_sweep<AUX>(P, S, b, L);
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
  const int b = blockIdx.x;
  n16::rts_bwd_sweep<HAS_FP>(P, S, U, G, ws, b, L);
  __syncthreads();   // the filter sweep reads back the hand-off records this wavefront has just written
  n16::filter_bwd_sweep<HAS_GQ>(P, S, G, ws, b, L);
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
