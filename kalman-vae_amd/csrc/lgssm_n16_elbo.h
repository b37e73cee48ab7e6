// lgssm_n16_elbo.h — the LGSSM terms of the sampled ELBO and their gradients for (n, m, p) = (16, 16, 2), one wavefront
// per (sequence, step) as in lgssm_elbo.h (same equations, cited there), but with the four 16-lane row-groups of the
// wavefront working on FOUR MATRICES AT ONCE, each with its rows on the lanes (lane i of a group owns row i, 16 registers):
//
//     group 0: Sigma_s[t]      group 1: Q_t (transition into t)      group 2: Q_{t+1}      group 3: Sigma0 (t = 0 only)
//
// One instruction stream factorises all four (right-looking Cholesky: the column broadcast is a DPP row_newbcast folded
// into the fma), forward- and back-substitutes the four residuals (wv, d_t, d_{t+1}, z_0 - mu0), and inverts the two
// factors the gradients need.  Nothing here is serial on one lane (the generic body solves its triangular systems on lane
// 0).  The 16x16x16 products of the Cholesky backward, g Sigma_s = sym(L^-T Phi L^-1), and of g Q_t run on the matrix cores
// in the C-layout of lgssm_n16.h after one hop through LDS.
//
// _safe_cholesky (kalman_filter.py:282-302): the probe launch resolves the whole-batch jitter level exactly as the generic
// probe does (first level 0..4 whose factorisation has all pivots > 0, 5 = diagonal fallback) and parks z_t = mu_t + L_t eps_t
// of level 0.  A raised level is the same factorisation of sym(X) + 1e-6 * 10^level I, and the diagonal fallback is the
// factorisation of diag(max(X_ii, 1e-6)) - so the main launches below take EVERY level (round 2 handed levels > 0 to the
// generic one-lane kernel: 3.3 ms instead of 0.4 at the configs[4] shard, and a learned Q[K,n,n] makes a raised level an
// ordinary event for the switching model): the matrix a row-group factorises is prepared according to its family's level, the
// instruction stream stays the same.  Only two things differ at level 5: the gradient reaches the un-clamped diagonal alone
// (sqrt(clamp(diag)) has no other inputs), and - at any level > 0 of Sigma_s - the parked z_t are redone first (elbo_zfix).
#pragma once
#include "lgssm_elbo.h"
#include "lgssm_n16.h"

#if !defined(KVAE_HOSTSIM) && !defined(KV_TPP)
namespace kvae {
namespace n16 {

struct alignas(16) ELds {
  float t[4][N * LD];   // one tile per row-group: rows-on-lanes -> C-layout
  Lds s;                // symmetrisation
};

// acc += (lane C of src's row) * f  - the DPP source is a different register than the accumulator
template <int C>
__device__ __forceinline__ void fmac_bcast_src(float &acc, float src, float f) {
  asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(f), "n"(C));
}
// two wait states between the VALU write of v and its next DPP read (the compiler cannot see into the asm above)
__device__ __forceinline__ void dpp_guard(float &v) { asm volatile("s_nop 1" : "+v"(v)); }

// Cholesky factor with its rows on the lanes
struct Chol {
  float l[N];    // row i of L (entries k <= i; garbage above the diagonal)
  float lm[N];   // -L[i][k] / L[k][k] for i > k, else 0: the unit-lower factor, negated and masked for the substitutions
  float rinv, ld;   // 1 / L[i][i], L[i][i]
  bool bad;         // a pivot was not > 0 (uniform within the row-group)
};
template <int K, int C>
__device__ __forceinline__ void chol_trail(float (&m)[N], float lk, float nl) {
  if constexpr (C < N) {
    fmac_bcast_src<C>(m[C], lk, nl);   // m[i][C] -= L[i][K] L[C][K]
    chol_trail<K, C + 1>(m, lk, nl);
  }
}
template <int K>
__device__ __forceinline__ void chol_step(float (&m)[N], Chol &c, int i) {
  const float d = bcast<K>(m[K]);
  c.bad |= !(d > 0.0f);
  const float ld = __builtin_amdgcn_sqrtf(d);
  const float rinv = frcp(ld);
  float lk = m[K] * rinv;                          // L[i][K] for i >= K
  c.l[K] = i == K ? ld : lk;
  c.lm[K] = i > K ? -(lk * rinv) : 0.0f;
  if (i == K) c.rinv = rinv, c.ld = ld;
  const float nl = -lk;
  dpp_guard(lk);
  chol_trail<K, K + 1>(m, lk, nl);
  if constexpr (K + 1 < N) chol_step<K + 1>(m, c, i);
}
__device__ __forceinline__ void cholesky_rows(float (&m)[N], Chol &c, int i) {
  c.bad = false, c.rinv = 0.f, c.ld = 1.f;
  chol_step<0>(m, c, i);
}

// y = Lu^{-1} x in place (Lu = unit-lower part of L): L^{-1} x = y * rinv
template <int K>
__device__ __forceinline__ void fsub_step(const Chol &c, float &x) {
  dpp_guard(x);
  fmac_bcast<K>(x, c.lm[K]);
  if constexpr (K + 2 < N) fsub_step<K + 1>(c, x);
}
// v = Lu^{-T} w in place
template <int K>
__device__ __forceinline__ void bsub_step(const Chol &c, float &v, int i) {
  const float s = row_sum(c.lm[K] * v);            // -sum_{i > K} Lu[i][K] v[i]
  v = i == K ? v + s : v;
  if constexpr (K > 0) bsub_step<K - 1>(c, v, i);
}

// X = L^{-1} with its rows on the lanes
template <int K, int C>
__device__ __forceinline__ void inv_cols(const Chol &c, float (&y)[N]) {
  if constexpr (C <= K) {
    fmac_bcast<K>(y[C], c.lm[K]);                  // y[i][C] -= Lu[i][K] y[K][C]
    inv_cols<K, C + 1>(c, y);
  }
}
template <int K>
__device__ __forceinline__ void inv_step(const Chol &c, float (&y)[N]) {
  // y[C] of the previous step (or the identity just selected into it) -> DPP read: the wait states must sit AFTER those writes,
  // so the guard names the registers (an operand-less s_nop let hipcc sink the initialisation of y[0] below it)
  asm volatile("s_nop 1" : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]), "+v"(y[4]), "+v"(y[5]), "+v"(y[6]), "+v"(y[7]), "+v"(y[8]),
               "+v"(y[9]), "+v"(y[10]), "+v"(y[11]), "+v"(y[12]), "+v"(y[13]), "+v"(y[14]), "+v"(y[15]));
  inv_cols<K, 0>(c, y);
  if constexpr (K + 2 < N) inv_step<K + 1>(c, y);
}
__device__ __forceinline__ void inverse_rows(const Chol &c, float (&x)[N], int i) {
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] = i == k ? 1.0f : 0.0f;
  inv_step<0>(c, x);
#pragma unroll
  for (int k = 0; k < N; ++k) x[k] *= c.rinv;      // L^{-1} = D^{-1} Lu^{-1}
}

// rows-on-lanes (this group's tile) -> LDS; upper triangle cleared
__device__ __forceinline__ void rows_to_tile(const float (&x)[N], float *tile, int i) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = 4 * q + r <= i ? x[4 * q + r] : 0.0f;
    *reinterpret_cast<f4 *>(&tile[i * LD + 4 * q]) = v;
  }
}
__device__ __forceinline__ f4 tile_c(const float *tile, int j, int g) {
  return f4{tile[(4 * g + 0) * LD + j], tile[(4 * g + 1) * LD + j], tile[(4 * g + 2) * LD + j], tile[(4 * g + 3) * LD + j]};
}

// row i of sym(X) + jitter I (rows_only: the lower triangle as it is - what torch.linalg.cholesky reads)
__device__ __forceinline__ void load_sym_rows(const float *X, float (&m)[N], int i, float jitter, bool rows_only) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4 v = *reinterpret_cast<const f4 *>(X + i * N + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * q + r;
      const float xt = X[c * N + i];
      m[c] = (rows_only ? v[r] : 0.5f * (v[r] + xt)) + (c == i ? jitter : 0.0f);
    }
  }
}
__device__ __forceinline__ void identity_rows(float (&m)[N], int i) {
#pragma unroll
  for (int c = 0; c < N; ++c) m[c] = c == i ? 1.0f : 0.0f;
}
__device__ __forceinline__ float diag_of_rows(const float (&m)[N], int i) {
  float d = 0.0f;
#pragma unroll
  for (int c = 0; c < N; ++c) d = c == i ? m[c] : d;
  return d;
}
// Row i of the matrix _safe_cholesky factorises at `level` (kalman_filter.py:286-302): sym(X) + 1e-6 * 10^level I for levels
// 0..4; at level 5 (no jitter repaired the batch) the factor is diag(sqrt(clamp(diag X, 1e-6))), i.e. the Cholesky factor of
// the diagonal matrix returned here.  `raw_diag` receives X_ii (the level-5 gradient only flows where X_ii >= 1e-6).
__device__ __forceinline__ void load_level_rows(const float *X, float (&m)[N], int i, int level, float &raw_diag) {
  load_sym_rows(X, m, i, 0.0f, false);
  raw_diag = diag_of_rows(m, i);
  const float jit = jitter_of_level(level < 5 ? level : 0);
#pragma unroll
  for (int c = 0; c < N; ++c) {
    const float lv = m[c] + (c == i ? jit : 0.0f);
    const float dg = c == i ? fmaxf(raw_diag, 1e-6f) : 0.0f;
    m[c] = level >= 5 ? dg : lv;
  }
}
// kernel families of the ELBO main launch, written to chol_levels[2] by the launch itself (tests assert which one ran)
enum : int { KV_ELBO_FAMILY_GENERIC = 0, KV_ELBO_FAMILY_TPP = 1, KV_ELBO_FAMILY_N16_STEP = 2, KV_ELBO_FAMILY_N16_FOUR = 3 };

// acc = sum_c M[i][c] v[c] with row i of M on the lane (16-byte loads) and v in L-layout
template <int C>
__device__ __forceinline__ void matvec_rows_acc(float &acc, const float (&row)[N], float vL) {
  if constexpr (C < N) {
    fmac_bcast_src<C>(acc, vL, row[C]);
    matvec_rows_acc<C + 1>(acc, row, vL);
  }
}
__device__ __forceinline__ float matvec_rows(const float *M, float vL, int i) {
  float row[N];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4 v = *reinterpret_cast<const f4 *>(M + i * N + 4 * q);
    row[4 * q] = v[0], row[4 * q + 1] = v[1], row[4 * q + 2] = v[2], row[4 * q + 3] = v[3];
  }
  float acc = 0.0f;
  dpp_guard(vL);
  matvec_rows_acc<0>(acc, row, vL);
  return acc;
}

// ---- probe: jitter levels + z_t = mu_t + chol(Sigma_s[t]) eps_t ------------------------------------------------------
template <int K>
__device__ __forceinline__ void lower_matvec_acc(float &acc, const Chol &c, float epsL, int i) {
  if constexpr (K < N) {
    fmac_bcast_src<K>(acc, epsL, K <= i ? c.l[K] : 0.0f);
    lower_matvec_acc<K + 1>(acc, c, epsL, i);
  }
}
__device__ __forceinline__ void elbo_probe(const kvae_lgssm_problem &P, const float *Sig_s, const float *mus, const float *eps,
                                           float *zst, int32_t *levels, int b, int t) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int64_t q = (int64_t)b * P.T + t;
  const bool q_shared = P.Q.sb == 0 && P.Q.st == 0;
  const bool probe_q = t >= 1 && (!q_shared || (b == 0 && t == 1));
  const bool valid = g == 0 || (g == 1 && probe_q);
  const float *X = g == 1 && probe_q ? stack_at(P.Q, b, t) : Sig_s + q * NN;
  float m0[N], m[N];
  load_sym_rows(X, m0, i, 0.0f, false);
  // Every row-group climbs the same ladder as probe_level() - jitter 1e-6 * 10^level - and keeps the first level at which
  // ITS matrix factorises with all pivots > 0 (5 = none does); the wavefront leaves the loop when every group has one.
  int lv = 5;
  bool done = false;
  for (int level = 0; level < 5; ++level) {
    const float jit = jitter_of_level(level);
#pragma unroll
    for (int k = 0; k < N; ++k) m[k] = valid ? m0[k] + (k == i ? jit : 0.0f) : (k == i ? 1.0f : 0.0f);
    Chol c;
    cholesky_rows(m, c, i);
    if (level == 0 && g == 0 && !c.bad) {   // z_t of level 0 for the main launch
      float epsL = eps[q * N + i];
      float acc = mus[q * N + i];
      dpp_guard(epsL);
      lower_matvec_acc<0>(acc, c, epsL, i);
      zst[q * N + i] = acc;
    }
    if (!done && !c.bad) lv = level, done = true;
    if (!__any(!done)) break;
  }
  if (i == 0 && g == 0 && lv > 0) atomic_max_i32(levels + 0, lv);
  if (i == 0 && g == 1 && lv > 0) atomic_max_i32(levels + 1, lv);
}

// ---- z_t again at the level the whole batch resolved to: only when Sigma_s left level 0 (the probe parked level-0 samples) ----
// Four steps per wavefront (group g: step t0 + g); every wavefront returns at its first instruction in the normal case.
__device__ __forceinline__ void elbo_zfix(const kvae_lgssm_problem &P, const float *Sig_s, const float *mus, const float *eps,
                                          float *zst, const int32_t *levels, int b, int t0) {
  const int lvS = levels[0];
  if (lvS == 0) return;
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int t = t0 + g;
  const bool valid = t < P.T;
  const int64_t q = (int64_t)b * P.T + (valid ? t : P.T - 1);
  float m[N], raw;
  load_level_rows(Sig_s + q * NN, m, i, lvS, raw);
  Chol c;
  cholesky_rows(m, c, i);
  float epsL = eps[q * N + i];
  float acc = mus[q * N + i];
  dpp_guard(epsL);
  lower_matvec_acc<0>(acc, c, epsL, i);
  if (valid) zst[q * N + i] = acc;
}

// ---- main: the four terms of step (b,t) and, with GRADS, every gradient of SUM(terms) (unit upstream) --------------
template <bool GRADS, bool HAS_GQ>
__device__ __forceinline__ void elbo_main(const kvae_lgssm_problem &P, const float *mus, const float *Sigs, const float *eps,
                                          float *terms, const int32_t *levels, const float *zst, float *g_mus, float *g_Sigs,
                                          const kvae_lgssm_input_grads &G, int b, int t, ELds &L) {
  const int lvS = levels[0], lvQ = levels[1];      // whole-batch levels of _safe_cholesky (probe launch): 0..4 jitter, 5 diagonal
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, T = P.T;
  const int64_t bT = (int64_t)b * T, q = bT + t;
  const bool has_prev = t >= 1, has_next = t + 1 < T;
  // ---- the four matrices, rows on lanes ----
  const bool valid = g == 0 || (g == 1 && has_prev) || (g == 2 && has_next && GRADS) || (g == 3 && t == 0);
  const float *X = Sigs + q * NN;
  if (g == 1 && has_prev) X = stack_at(P.Q, b, t);
  if (g == 2 && has_next) X = stack_at(P.Q, b, t + 1);
  if (g == 3 && t == 0) X = P.Sigma0 + (int64_t)b * P.Sigma0_sb;
  float m[N], raw_diag;
  if (g == 3) {                                    // Sigma0 as MultivariateNormal(covariance_matrix=...) takes it: no jitter
    load_sym_rows(X, m, i, 0.0f, true);
    raw_diag = 1.0f;
  } else {
    load_level_rows(X, m, i, g == 0 ? lvS : lvQ, raw_diag);
  }
  if (!valid) identity_rows(m, i);
  Chol c;
  cholesky_rows(m, c, i);
  if (blockIdx.x == 0 && lane == 0) const_cast<int32_t *>(levels)[2] = KV_ELBO_FAMILY_N16_STEP;
  // ---- z, operands, residuals: wv (g0), d_t (g1), d_{t+1} (g2), z_0 - mu0 (g3) ----
  const float zt = zst[q * N + i];
  const float zp = has_prev ? zst[(q - 1) * N + i] : 0.0f, zn = has_next ? zst[(q + 1) * N + i] : 0.0f;
  const int ts = g == 2 && has_next ? t + 1 : t;           // the step whose A, B, u this group multiplies
  const float vecL = g == 2 ? zt : zp;
  const float uL = P.U[(bT + ts) * N + i];
  const float Az = matvec_rows(stack_at(P.A, b, ts), vecL, i);
  const float Bu = matvec_rows(stack_at(P.Bm, b, ts), uL, i);
  float rhs = zt - mus[q * N + i];
  if (g == 1) rhs = zt - (Az + Bu);
  if (g == 2) rhs = zn - (Az + Bu);
  if (g == 3) rhs = zt - P.mu0[(int64_t)b * P.mu0_sb + i];
  if (!valid) rhs = 0.0f;
  // ---- L^{-1} rhs, quadratic forms, log-determinants ----
  float y = rhs;
  fsub_step<0>(c, y);
  const float xf = y * c.rinv;
  const float quad = row_sum(xf * xf), logdet = row_sum(__logf(c.ld));
  // emission (p = 2): e = y - C z, R factorised without jitter
  const float *C = stack_at(P.C, b, t);
  const float Cl0 = C[i], Cl1 = C[N + i];
  const float e0 = P.Y[q * 2] - row_sum(Cl0 * zt), e1 = P.Y[q * 2 + 1] - row_sum(Cl1 * zt);
  const float l00 = __builtin_amdgcn_sqrtf(P.R[0]), l10 = P.R[2] / l00, l11 = __builtin_amdgcn_sqrtf(P.R[3] - l10 * l10);
  const float w0 = e0 / l00, w1 = (e1 - l10 * w0) / l11;
  const float qe1 = w1 / l11, qe0 = (w0 - l10 * qe1) / l00;      // R^{-1} e
  const float mkr = *mask_addr(P, b, t);
  const float mk = P.mask ? mkr : 1.0f;
  const float gauss = -0.5f * (N * KV_LOG2PI + quad) - logdet;   // log N(rhs; 0, L L^T) of this group
  if (i == 0) {
    if (g == 0) {
      terms[q * 4 + 3] = -gauss;                                 // entropy
      terms[q * 4 + 1] = mk * (-0.5f * (2 * KV_LOG2PI + (w0 * w0 + w1 * w1)) - (__logf(l00) + __logf(l11)));
    }
    if (g == 1) terms[q * 4 + 0] = has_prev ? gauss : 0.0f;      // transition
    if (g == 3) terms[q * 4 + 2] = t == 0 ? gauss : 0.0f;        // init
  }
  if constexpr (GRADS) {
    // ---- (L L^T)^{-1} rhs in every group: v_t (g1), v_{t+1} (g2), Sigma0^{-1}(z0 - mu0) (g3) ----
    float v = xf * c.rinv;
    bsub_step<N - 1>(c, v, i);
    const float vtL = has_prev ? __shfl(v, 16 + i, 64) : 0.0f;   // L-layout copies in every group
    const float viL = t == 0 ? __shfl(v, 48 + i, 64) : 0.0f;
    f4 vtW, vnW;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      vtW[r] = has_prev ? __shfl(v, 16 + 4 * g + r, 64) : 0.0f;
      vnW[r] = has_next ? __shfl(v, 32 + 4 * g + r, 64) : 0.0f;
    }
    // gz_t = -v_t + A_{t+1}^T v_{t+1} + mask C^T R^{-1} e - Sigma0^{-1}(z_0 - mu0)
    const f4 AnC = load_c(stack_at(P.A, b, has_next ? t + 1 : t), i, g);
    const float gz = -vtL + mtv(AnC, vnW) + mk * (Cl0 * qe0 + Cl1 * qe1) - viL;
    g_mus[q * N + i] = gz;
    // input gradients of step t
    float *gC = gstack_at(G.gC, b, t);
    gC[i] = mk * qe0 * zt, gC[N + i] = mk * qe1 * zt;
    G.gY[q * 2 + (i & 1)] = -mk * ((i & 1) ? qe1 : qe0);
    const f4 zpW = has_prev ? load_w(zst + (q - 1) * N, g) : zero4();
    store_rows(gstack_at(G.gA, b, t), vtL * zpW, i, g);          // gA_t = v_t z_{t-1}^T (row i, columns 4g..4g+3)
    store_rows(gstack_at(G.gB, b, t), vtL * load_w(P.U + q * N, g), i, g);
    if (G.gU) G.gU[q * N + i] = mtv(load_c(stack_at(P.Bm, b, t), i, g), vtW);   // B_t^T v_t
    // ---- Cholesky backward into Sigma_s: sym(L^{-T} Phi L^{-1}), Phi = strict_tril(a eps^T) + diag((a eps + 1)/2), a = L^T gz
    __syncthreads();
    rows_to_tile(c.l, L.t[g], i);
    __syncthreads();
    const f4 Lc = tile_c(L.t[0], i, g);                          // C-layout of L_s
    const float aL = mtv(Lc, l2w(gz, lane));
    float xr[N];
    inverse_rows(c, xr, i);                                      // L^{-1} of every group
    __syncthreads();
    rows_to_tile(xr, L.t[g], i);
    __syncthreads();
    const float epsL = eps[q * N + i];
    if (lvS >= 5) {   // (uniform) diagonal fallback: L = diag(sqrt(clamp(diag Sigma_s))), the gradient reaches the un-clamped diagonal only
      const float ls = __shfl(c.ld, i, 64), rawS = __shfl(raw_diag, i, 64);
      const float dv = rawS >= 1e-6f ? (gz * epsL + 1.0f / ls) / (2.0f * ls) : 0.0f;
      f4 dg;
#pragma unroll
      for (int r = 0; r < 4; ++r) dg[r] = 4 * g + r == i ? dv : 0.0f;
      store_rows(g_Sigs + q * NN, dg, i, g);
    } else {
      const f4 Xs = tile_c(L.t[0], i, g);
      const f4 epsW = load_w(eps + q * N, g);
      f4 Pht;                                                      // C-layout of Phi^T: [4g+r][i] = Phi[i][4g+r]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 4 * g + r;
        Pht[r] = col < i ? aL * epsW[r] : (col == i ? 0.5f * (aL * epsL + 1.0f) : 0.0f);
      }
      const f4 PX = mtn(Pht, Xs);                                  // Phi L^{-1}
      const f4 S = mtn(Xs, PX);                                    // L^{-T} Phi L^{-1}
      store_rows(g_Sigs + q * NN, symmetrise(S, L.s, i, g), i, g);
    }
    if constexpr (HAS_GQ) {
      f4 gq = zero4();
      if (lvQ >= 5) {   // (uniform) log N(d; 0, diag l^2): d/dl_i = -1/l_i + d_i^2 / l_i^3, dl_i/dq_ii = 1/(2 l_i) where q_ii >= 1e-6
        const float lq = __shfl(c.ld, 16 + i, 64), rawQ = __shfl(raw_diag, 16 + i, 64), dq = __shfl(rhs, 16 + i, 64);
        const float dv = (has_prev && rawQ >= 1e-6f) ? (-1.0f / lq + dq * dq / (lq * lq * lq)) / (2.0f * lq) : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) gq[r] = 4 * g + r == i ? dv : 0.0f;
      } else {          // gQ_t = 1/2 v v^T - 1/2 (L_Q L_Q^T)^{-1}
        const f4 Xq = tile_c(L.t[1], i, g);
        const f4 Qi = mtn(Xq, Xq);
        gq = has_prev ? 0.5f * (vtW * vtL - Qi) : zero4();
      }
      store_c(gstack_at(G.gQ, b, t), gq, i, g);
    }
  }
}

// ---- shared Q (lstm dynamics: Q is one matrix for the whole batch): FOUR STEPS per wavefront ---------------------------------
// With a shared Q three of the four row-groups of elbo_probe / elbo_main factorise the same matrix again in every wavefront, or
// idle.  Here group g works on step t0 + g: the launch has a quarter of the wavefronts, and the kernels are bound by VALU
// issue.  Q is factorised once per wavefront (all groups, redundantly: X = L_Q^-1 and X^T with their rows on the lanes, so that
// v = Q^-1 d is two 16-FMA mat-vecs and the quadratic form is |X d|^2 exactly as torch's MultivariateNormal computes it);
// v_{t+1} is recomputed by the group of step t (its A_{t+1}, B_{t+1} rows come from L1 / L2: the neighbouring group reads the
// same lines); the Cholesky backward runs its two MFMA products once per group.
struct alignas(16) ELds4 {
  float lt[4][N * LD];   // L of every group's Sigma_s (rows-on-lanes -> C-layout); first used to transpose L_Q^-1
  float xt[4][N * LD];   // L^-1
  float gz[4][N];        // d ELBO / d z_t of every group
  Lds s;
};

__device__ __forceinline__ void elbo_probe4(const kvae_lgssm_problem &P, const float *Sig_s, const float *mus, const float *eps,
                                            float *zst, int32_t *levels, int b, int t0, bool probe_q) {
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4;
  const int t = t0 + g;
  const bool valid = t < P.T;
  const int64_t q = (int64_t)b * P.T + (valid ? t : P.T - 1);
  float m0[N], m[N];
  load_sym_rows(Sig_s + q * NN, m0, i, 0.0f, false);
  int lv = 5;
  bool done = false;
  for (int level = 0; level < 5; ++level) {
    const float jit = jitter_of_level(level);
#pragma unroll
    for (int k = 0; k < N; ++k) m[k] = valid ? m0[k] + (k == i ? jit : 0.0f) : (k == i ? 1.0f : 0.0f);
    Chol c;
    cholesky_rows(m, c, i);
    if (level == 0 && valid && !c.bad) {   // z_t of level 0 for the main launch
      float epsL = eps[q * N + i];
      float acc = mus[q * N + i];
      dpp_guard(epsL);
      lower_matvec_acc<0>(acc, c, epsL, i);
      zst[q * N + i] = acc;
    }
    if (!done && !c.bad) lv = level, done = true;
    if (!__any(!done)) break;
  }
  if (i == 0 && valid && lv > 0) atomic_max_i32(levels + 0, lv);
  if (probe_q) {                           // one wavefront of the launch: the ladder of the shared Q
    load_sym_rows(P.Q.ptr, m0, i, 0.0f, false);
    int lq = 5;
    for (int level = 0; level < 5; ++level) {
      const float jit = jitter_of_level(level);
#pragma unroll
      for (int k = 0; k < N; ++k) m[k] = m0[k] + (k == i ? jit : 0.0f);
      Chol c;
      cholesky_rows(m, c, i);
      if (!c.bad) { lq = level; break; }   // the four groups hold the same matrix: uniform
    }
    if (lane == 0 && lq > 0) atomic_max_i32(levels + 1, lq);
  }
}

// acc += sum_k row[k] * (lane k of vL's row)
__device__ __forceinline__ float matvec_reg(const float (&row)[N], float vL) {
  float acc = 0.0f;
  dpp_guard(vL);
  matvec_rows_acc<0>(acc, row, vL);
  return acc;
}
// column i of a row-major 16x16 matrix (row i of its transpose)
__device__ __forceinline__ void load_col(const float *M, float (&col)[N], int i) {
#pragma unroll
  for (int k = 0; k < N; ++k) col[k] = M[k * N + i];
}

template <bool GRADS>
__device__ __forceinline__ void elbo_main4(const kvae_lgssm_problem &P, const float *mus, const float *Sigs, const float *eps,
                                           float *terms, const int32_t *levels, const float *zst, float *g_mus, float *g_Sigs,
                                           const kvae_lgssm_input_grads &G, int b, int t0, ELds4 &L) {
  const int lvS = levels[0], lvQ = levels[1];      // whole-batch levels of _safe_cholesky (probe launch): 0..4 jitter, 5 diagonal
  const int lane = threadIdx.x & 63, i = lane & 15, g = lane >> 4, T = P.T;
  if (blockIdx.x == 0 && lane == 0) const_cast<int32_t *>(levels)[2] = KV_ELBO_FAMILY_N16_FOUR;
  const int tr = t0 + g;
  const bool valid = tr < T;
  const int t = valid ? tr : T - 1;                // clamped: invalid groups recompute the last step and store nothing
  const int64_t bT = (int64_t)b * T, q = bT + t;
  const bool has_prev = t >= 1, has_next = t + 1 < T;
  // ---- the shared Q: X = (chol(Q + 1e-6 I))^-1 and X^T, rows on lanes; log-determinant ----
  float xq[N], xqt[N], logdetQ;
  {
    float mq[N], rawq;
    load_level_rows(P.Q.ptr, mq, i, lvQ, rawq);
    Chol cq;
    cholesky_rows(mq, cq, i);
    inverse_rows(cq, xq, i);
    logdetQ = row_sum(__logf(cq.ld));
    __syncthreads();
    rows_to_tile(xq, L.lt[g], i);                  // upper triangle cleared: X is lower triangular
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) xqt[k] = L.lt[g][k * LD + i];
#pragma unroll
    for (int k = 0; k < N; ++k) xq[k] = k <= i ? xq[k] : 0.0f;
  }
  // ---- Sigma_s[t] of this group ----
  float m[N], raw_diag;
  load_level_rows(Sigs + q * NN, m, i, lvS, raw_diag);
  Chol c;
  cholesky_rows(m, c, i);
  const float zt = zst[q * N + i];
  const float zp = has_prev ? zst[(q - 1) * N + i] : 0.0f, zn = has_next ? zst[(q + 1) * N + i] : 0.0f;
  // entropy: log N(z_t; mu_s, L L^T)
  float y = zt - mus[q * N + i];
  fsub_step<0>(c, y);
  const float xf = y * c.rinv;
  const float gaussS = -0.5f * (N * KV_LOG2PI + row_sum(xf * xf)) - row_sum(__logf(c.ld));
  // transition into t: d_t = z_t - A_t z_{t-1} - B_t u_t, w = X d, v_t = X^T w
  const float Az = matvec_rows(stack_at(P.A, b, t), zp, i);
  const float Bu = matvec_rows(stack_at(P.Bm, b, t), P.U[q * N + i], i);
  const float dt = has_prev ? zt - (Az + Bu) : 0.0f;
  const float wt = matvec_reg(xq, dt);
  const float gaussT = -0.5f * (N * KV_LOG2PI + row_sum(wt * wt)) - logdetQ;
  // emission (p = 2): e = y - C z, R factorised without jitter
  const float *C = stack_at(P.C, b, t);
  const float Cl0 = C[i], Cl1 = C[N + i];
  const float e0 = P.Y[q * 2] - row_sum(Cl0 * zt), e1 = P.Y[q * 2 + 1] - row_sum(Cl1 * zt);
  const float l00 = __builtin_amdgcn_sqrtf(P.R[0]), l10 = P.R[2] / l00, l11 = __builtin_amdgcn_sqrtf(P.R[3] - l10 * l10);
  const float w0 = e0 / l00, w1 = (e1 - l10 * w0) / l11;
  const float qe1 = w1 / l11, qe0 = (w0 - l10 * qe1) / l00;      // R^{-1} e
  const float mkr = *mask_addr(P, b, t);
  const float mk = P.mask ? mkr : 1.0f;
  // init (the wavefront that holds t = 0): log N(z_0; mu0, Sigma0), Sigma0 factorised as MultivariateNormal does (no jitter)
  float gaussI = 0.0f, viL = 0.0f;
  if (t0 == 0) {
    float m0[N];
    load_sym_rows(P.Sigma0 + (int64_t)b * P.Sigma0_sb, m0, i, 0.0f, true);
    Chol c0;
    cholesky_rows(m0, c0, i);
    float y0 = zst[bT * N + i] - P.mu0[(int64_t)b * P.mu0_sb + i];
    fsub_step<0>(c0, y0);
    const float x0 = y0 * c0.rinv;
    gaussI = -0.5f * (N * KV_LOG2PI + row_sum(x0 * x0)) - row_sum(__logf(c0.ld));
    float v0 = x0 * c0.rinv;
    bsub_step<N - 1>(c0, v0, i);
    viL = t == 0 ? v0 : 0.0f;                       // every group computed step 0's; only its own group uses it
  }
  if (i == 0 && valid) {
    terms[q * 4 + 0] = has_prev ? gaussT : 0.0f;
    terms[q * 4 + 1] = mk * (-0.5f * (2 * KV_LOG2PI + (w0 * w0 + w1 * w1)) - (__logf(l00) + __logf(l11)));
    terms[q * 4 + 2] = t == 0 ? gaussI : 0.0f;
    terms[q * 4 + 3] = -gaussS;
  }
  if constexpr (GRADS) {
    const float vt = matvec_reg(xqt, wt);                         // Q^-1 d_t (0 at t = 0)
    // v_{t+1}, recomputed here (the next group / wavefront computes it as its own v_t)
    const float *An = stack_at(P.A, b, has_next ? t + 1 : t), *Bn = stack_at(P.Bm, b, has_next ? t + 1 : t);
    const float Azn = matvec_rows(An, zt, i);
    const float Bun = matvec_rows(Bn, P.U[(q + (has_next ? 1 : 0)) * N + i], i);
    const float dn = has_next ? zn - (Azn + Bun) : 0.0f;
    const float vn = matvec_reg(xqt, matvec_reg(xq, dn));
    // gz_t = -v_t + A_{t+1}^T v_{t+1} + mask C^T R^{-1} e - Sigma0^{-1}(z_0 - mu0)
    float col[N];
    load_col(An, col, i);
    const float gz = -vt + matvec_reg(col, vn) + mk * (Cl0 * qe0 + Cl1 * qe1) - viL;
    if (valid) {
      g_mus[q * N + i] = gz;
      float *gC = gstack_at(G.gC, b, t);
      gC[i] = mk * qe0 * zt, gC[N + i] = mk * qe1 * zt;
      if (i < 2) G.gY[q * 2 + i] = -mk * (i ? qe1 : qe0);
      // gA_t = v_t z_{t-1}^T, gB_t = v_t u_t^T: row i on lane i
      float *gA = gstack_at(G.gA, b, t) + i * N, *gB = gstack_at(G.gB, b, t) + i * N;
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const f4 zq = has_prev ? load_w(zst + (q - 1) * N, k4) : zero4();
        *reinterpret_cast<f4 *>(gA + 4 * k4) = vt * zq;
        *reinterpret_cast<f4 *>(gB + 4 * k4) = vt * load_w(P.U + q * N, k4);
      }
      if (G.gU) {
        load_col(stack_at(P.Bm, b, t), col, i);
        G.gU[q * N + i] = matvec_reg(col, vt);                    // B_t^T v_t
      }
    }
    // ---- Cholesky backward into Sigma_s of every group: sym(L^-T Phi L^-1) on the matrix cores, one group after the other ----
    float xr[N];
    inverse_rows(c, xr, i);
    __syncthreads();
    rows_to_tile(c.l, L.lt[g], i);
    rows_to_tile(xr, L.xt[g], i);
    L.gz[g][i] = gz;
    __syncthreads();
    if (lvS >= 5) {   // (uniform) diagonal fallback: the gradient reaches the un-clamped diagonal of this group's Sigma_s only
      const float dv = raw_diag >= 1e-6f ? (gz * eps[q * N + i] + 1.0f / c.ld) / (2.0f * c.ld) : 0.0f;
      if (valid) {
        float *gS = g_Sigs + q * NN + i * N;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          f4 dg;
#pragma unroll
          for (int r = 0; r < 4; ++r) dg[r] = 4 * k4 + r == i ? dv : 0.0f;
          *reinterpret_cast<f4 *>(gS + 4 * k4) = dg;
        }
      }
      return;
    }
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) {
      if (t0 + gg >= T) break;                                    // wave-uniform
      const int64_t qg = bT + t0 + gg;
      const f4 Lc = tile_c(L.lt[gg], i, g);                       // C-layout of L_s
      const float aL = mtv(Lc, load_w(L.gz[gg], g));              // a = L^T gz
      const f4 Xs = tile_c(L.xt[gg], i, g);
      const f4 epsW = load_w(eps + qg * N, g);
      const float epsL = eps[qg * N + i];
      f4 Pht;                                                      // C-layout of Phi^T: [4g+r][i] = Phi[i][4g+r]
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int cc = 4 * g + r;
        Pht[r] = cc < i ? aL * epsW[r] : (cc == i ? 0.5f * (aL * epsL + 1.0f) : 0.0f);
      }
      const f4 PX = mtn(Pht, Xs);                                  // Phi L^{-1}
      const f4 S = mtn(Xs, PX);                                    // L^{-T} Phi L^{-1}
      store_rows(g_Sigs + qg * NN, symmetrise(S, L.s, i, g), i, g);
    }
  }
}


// ---- the kernels' bodies (kvae_lgssm_n16.hip wraps them in __global__ functions with the tiles in LDS).  One wavefront per step (per-step Q) or per four consecutive steps (shared Q); wavefront w of the grid
// works on the w-th unit of ITS XCD's contiguous share, so that a step and its successor - which read each other's operands -
// sit in the same L2.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned wg, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7, xcd = wg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
}
__device__ __forceinline__ void elbo_probe_wave(const kvae_lgssm_problem &P, const float *Sig_s, const float *mus, const float *eps,
                                                float *zst, int32_t *levels) {
  const unsigned w = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = w / P.T, t = w - b * P.T;
  elbo_probe(P, Sig_s, mus, eps, zst, levels, b, t);
}
template <bool GRADS, bool HAS_GQ>
__device__ __forceinline__ void elbo_wave(const kvae_lgssm_problem &P, const float *mus, const float *Sigs, const float *eps,
                                          float *terms, const int32_t *levels, const float *zst, float *g_mus, float *g_Sigs,
                                          const kvae_lgssm_input_grads &G, ELds &L) {
  const unsigned w = xcd_contiguous(blockIdx.x, gridDim.x);
  const int b = w / P.T, t = w - b * P.T;
  elbo_main<GRADS, HAS_GQ>(P, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, G, b, t, L);
}
__device__ __forceinline__ void elbo_probe4_wave(const kvae_lgssm_problem &P, const float *Sig_s, const float *mus, const float *eps,
                                                 float *zst, int32_t *levels) {
  const unsigned w = xcd_contiguous(blockIdx.x, gridDim.x);
  const int nq = (P.T + 3) >> 2, b = w / nq, t0 = 4 * (w - b * nq);
  elbo_probe4(P, Sig_s, mus, eps, zst, levels, b, t0, w == 0 && P.T >= 2);
}
template <bool GRADS>
__device__ __forceinline__ void elbo4_wave(const kvae_lgssm_problem &P, const float *mus, const float *Sigs, const float *eps,
                                           float *terms, const int32_t *levels, const float *zst, float *g_mus, float *g_Sigs,
                                           const kvae_lgssm_input_grads &G, ELds4 &L) {
  const unsigned w = xcd_contiguous(blockIdx.x, gridDim.x);
  const int nq = (P.T + 3) >> 2, b = w / nq, t0 = 4 * (w - b * nq);
  elbo_main4<GRADS>(P, mus, Sigs, eps, terms, levels, zst, g_mus, g_Sigs, G, b, t0, L);
}
__device__ __forceinline__ void elbo_zfix_wave(const kvae_lgssm_problem &P, const float *Sig_s, const float *mus, const float *eps,
                                               float *zst, const int32_t *levels) {
  const unsigned w = xcd_contiguous(blockIdx.x, gridDim.x);
  const int nq = (P.T + 3) >> 2, b = w / nq, t0 = 4 * (w - b * nq);
  elbo_zfix(P, Sig_s, mus, eps, zst, levels, b, t0);
}
}  // namespace n16
}  // namespace kvae
#endif
