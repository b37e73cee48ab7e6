// kvae_lgssm_rnn.hip — parameter gradients of the alpha-network recurrences and their linear heads (rnn_wgrad.h,
// small_linear.h): what round 2 left to rocBLAS inside the captured LGSSM chain.  Public entry points: kvae_lgssm.hip.
#include <hip/hip_runtime.h>

#include "../../include/kvae_lgssm.h"
#include "rnn_wgrad.h"
#include "small_linear.h"

using namespace kvae;

static_assert(sizeof(kvae_wgrad_problem) == sizeof(WgradProblem), "C-ABI struct and kernel struct must agree");

static inline int wg_cpad(const kvae_wgrad_problem &p) { return (p.H + p.I + (p.bias ? 1 : 0) + 15) / 16 * 16; }

extern "C" int64_t kvae_rnn_wgrad_ws_floats(const kvae_wgrad_problem *probs, int32_t n) {
  int64_t per = 0;
  for (int i = 0; i < n; ++i) {
    const int64_t e = (int64_t)probs[i].R * wg_cpad(probs[i]);
    per = e > per ? e : per;
  }
  return per * WG_MAX_CHUNKS * n;
}

// returns 0, or a KVAE_ERR_* code for bad arguments (nothing launched)
extern "C" int kvae_rnn_launch_wgrad(const kvae_wgrad_problem *probs, int32_t n, float *ws, hipStream_t s) {
  WgradBatch batch;
  batch.n = n;
  int max_rt = 0, max_groups = 0;
  int64_t per = 0, max_n = 0, max_elems = 0;
  for (int i = 0; i < n; ++i) {
    const kvae_wgrad_problem &p = probs[i];
    const int C = p.H + p.I + (p.bias ? 1 : 0);
    if (!p.d || (p.H > 0 && !p.h) || (p.I > 0 && !p.x)) return KVAE_ERR_NULL;
    if (p.N < 1 || p.R < 1 || p.R > 16 * WG_MAX_ROW_TILES || p.H < 0 || p.I < 0 || C < 1 || C > 256 || p.T < 1 || p.shift < -1 ||
        p.shift > 1 || p.N % p.T != 0)
      return KVAE_ERR_ARG;
    WgradProblem &q = batch.p[i];
    q.d = p.d, q.h = p.h, q.x = p.x, q.g_wh = p.g_wh, q.g_wx = p.g_wx, q.g_b = p.g_b;
    q.d_stride = p.d_stride, q.h_stride = p.h_stride, q.x_stride = p.x_stride, q.N = p.N;
    q.R = p.R, q.H = p.H, q.I = p.I, q.bias = p.bias, q.T = p.T, q.shift = p.shift;
    const int rt = (p.R + 15) / 16, ct = wg_cpad(p) / 16;
    max_rt = rt > max_rt ? rt : max_rt;
    max_groups = (ct + 3) / 4 > max_groups ? (ct + 3) / 4 : max_groups;
    const int64_t e = (int64_t)p.R * wg_cpad(p);
    per = e > per ? e : per;
    max_n = p.N > max_n ? p.N : max_n;
    max_elems = (int64_t)p.R * C > max_elems ? (int64_t)p.R * C : max_elems;
  }
  // split-K: about two workgroups per CU over all problems and column groups, at least one 32-row stage each
  int chunks = (int)((max_n + 31) / 32);
  const int fill = 512 / (n * max_groups);
  chunks = chunks > fill ? fill : chunks;
  chunks = chunks > WG_MAX_CHUNKS ? WG_MAX_CHUNKS : (chunks < 1 ? 1 : chunks);
  const int64_t stride = per * WG_MAX_CHUNKS;
  const dim3 grid((unsigned)chunks, (unsigned)max_groups, (unsigned)n), block(256);
  if (max_rt <= 1) k_rnn_wgrad_partial<1><<<grid, block, 0, s>>>(batch, ws, stride, chunks);
  else if (max_rt <= 4) k_rnn_wgrad_partial<4><<<grid, block, 0, s>>>(batch, ws, stride, chunks);
  else if (max_rt <= 10) k_rnn_wgrad_partial<10><<<grid, block, 0, s>>>(batch, ws, stride, chunks);
  else if (max_rt <= 13) k_rnn_wgrad_partial<13><<<grid, block, 0, s>>>(batch, ws, stride, chunks);
  else k_rnn_wgrad_partial<16><<<grid, block, 0, s>>>(batch, ws, stride, chunks);
  k_rnn_wgrad_final<<<dim3((unsigned)((max_elems + 31) / 32), (unsigned)n), block, 0, s>>>(batch, ws, stride, chunks);
  return KVAE_OK;
}

extern "C" int kvae_rnn_launch_linear_fwd(const float *x, int64_t xs, int64_t N, int F, const float *W, const float *b, int O,
                                          int softmax, float *y, hipStream_t s) {
  const int rows = sl_rows_per_block(F, O, N);
  const size_t lds = sizeof(float) * ((size_t)O * (F + 1) + (size_t)rows * (F + 1));
  const dim3 grid((unsigned)((N + rows - 1) / rows));
  if (softmax)
    k_linear_softmax_fwd<<<grid, dim3(256), lds, s>>>(x, xs, N, F, W, b, O, y, rows);
  else
    k_linear_fwd<<<grid, dim3(256), lds, s>>>(x, xs, N, F, W, b, O, y, rows);
  return KVAE_OK;
}

extern "C" int kvae_rnn_launch_linear_bwd_input(const float *g, const float *y, int64_t N, int F, const float *W, int O,
                                                float *g_logit, float *dx, int64_t dxs, hipStream_t s) {
  const size_t lds = sizeof(float) * O * (F + 1);
  const int64_t threads = N * ((F + 3) / 4);
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (y)
    k_linear_softmax_bwd_input<<<grid, dim3(256), lds, s>>>(g, y, N, F, W, O, g_logit, dx, dxs);
  else
    k_linear_bwd_input<<<grid, dim3(256), lds, s>>>(g, N, F, W, O, dx, dxs);
  return KVAE_OK;
}
