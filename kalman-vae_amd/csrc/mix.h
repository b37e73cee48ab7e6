// mix.h — mixture-of-K dynamics: per-step matrices as alpha-weighted sums of K base matrices
// (reference: dyn_param.py:58-60, switch_dyn_param.py:82-84) and the reverse mode.
// Element-wise / reduction work over [rows = B*T, E] records; HBM-bound, coalesced along E.
#pragma once
#include <stdint.h>

#include "lgssm_vm.h"

#define KVAE_MIX_ROWS_PER_BLOCK 16
#define KVAE_MAX_K 16

namespace kvae {

// out[r, e] = sum_k alpha[r,k] base[k,e]
KV_DEV void mix_fwd_elem(const float *alpha, const float *base, float *out, int64_t idx, int K, int E) {
  const int64_t r = idx / E;
  const int e = (int)(idx - r * E);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc = fmaf(alpha[r * K + k], base[k * E + e], acc);
  out[idx] = acc;
}

// g_alpha[r,k] (+)= sum_e g_out[r,e] base[k,e]
KV_DEV void mix_bwd_alpha_elem(const float *base, const float *g_out, float *g_alpha, int64_t idx, int K, int E,
                               int accumulate) {
  const int64_t r = idx / K;
  const int k = (int)(idx - r * K);
  float acc = 0.f;
  for (int e = 0; e < E; ++e) acc = fmaf(g_out[r * E + e], base[k * E + e], acc);
  g_alpha[idx] = accumulate ? g_alpha[idx] + acc : acc;
}

// partials[blk, k, e] = sum_{r in slab blk} alpha[r,k] g_out[r,e]
KV_DEV void mix_bwd_partial_elem(const float *alpha, const float *g_out, float *partials, int64_t blk, int e,
                                 int64_t rows, int K, int E) {
  float acc[KVAE_MAX_K];
  for (int k = 0; k < KVAE_MAX_K; ++k) acc[k] = 0.f;
  const int64_t r0 = blk * KVAE_MIX_ROWS_PER_BLOCK;
  const int64_t r1 = (r0 + KVAE_MIX_ROWS_PER_BLOCK < rows) ? r0 + KVAE_MIX_ROWS_PER_BLOCK : rows;
  float g[KVAE_MIX_ROWS_PER_BLOCK];  // issue every row's load before the first use (independent, latency overlapped)
  KV_UNROLL
  for (int i = 0; i < KVAE_MIX_ROWS_PER_BLOCK; ++i) g[i] = (r0 + i < r1) ? g_out[(r0 + i) * E + e] : 0.0f;
  KV_UNROLL
  for (int i = 0; i < KVAE_MIX_ROWS_PER_BLOCK; ++i) {
    const int64_t r = (r0 + i < r1) ? r0 + i : r0;
    for (int k = 0; k < KVAE_MAX_K; ++k)
      if (k < K) acc[k] = fmaf(alpha[r * K + k], g[i], acc[k]);
  }
  for (int k = 0; k < KVAE_MAX_K; ++k)
    if (k < K) partials[(blk * K + k) * E + e] = acc[k];
}

// g_base[k,e] = sum_blk partials[blk,k,e]   (fixed order: bit-reproducible)
KV_DEV void mix_bwd_final_elem(const float *partials, float *g_base, int idx, int64_t nblk, int KE) {
  float acc = 0.f;
  for (int64_t blk = 0; blk < nblk; ++blk) acc += partials[blk * KE + idx];
  g_base[idx] = acc;
}

}  // namespace kvae
