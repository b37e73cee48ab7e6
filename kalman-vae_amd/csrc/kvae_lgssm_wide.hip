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
