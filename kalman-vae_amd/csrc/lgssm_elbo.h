// lgssm_elbo.h — the LGSSM terms of the sampled ELBO and their gradients, ONE (sequence, step)
// per wavefront (no recursion in t: every (b,t) only needs z_{t-1}, z_t, z_{t+1}).
// Reference: KalmanFilter.elbo and _safe_cholesky (kalman_filter.py:282-302, 305-401) with
// torch.distributions.MultivariateNormal.log_prob / rsample semantics:
//     log N(x; 0, L L^T) = -1/2 (n log 2pi + |L^{-1} x|^2) - sum log diag L,   z = mu + L eps.
//
//   elbo_probe_body : per (b,t), smallest jitter level at which chol(sym(Sigma_s)+j I) and
//                     chol(sym(Q_t)+j I) succeed; atomicMax into levels[0..1]
//   elbo_body       : per (b,t), the four terms and (optionally) unit-upstream gradients
#pragma once
#include "lgssm_fwd.h"

namespace kvae {

#define KV_LOG2PI 1.8378770664093453f

template <class D>
struct ElboLds {
  static constexpr int N = D::NMAX, M = D::MMAX, P = D::PMAX;
  float sym[N * N];            // sym(X) + jitter I, the matrix being factorised
  float Ls[N * N];             // chol of Sigma_s[t]
  float LQ[N * N], LQn[N * N]; // chol of Q_t and Q_{t+1}
  float LR[P * P], L0[N * N], R[P * P];
  float z[3][N];               // z_{t-1}, z_t, z_{t+1}
  float eps[N], mu[N];
  float A[N * N], Bm[N * M], C[P * N], u[M], y[P], An[N * N], Bn[N * M], un[M];
  float dt[N], dn[N], vt[N], vn[N]; // transition residuals and Q^{-1} d for t and t+1
  float em[P], qe[P];          // emission residual and R^{-1} e
  float di[N], vi[N];          // init residual and Sigma0^{-1} (z0 - mu0)
  float wv[N];                 // L^{-1}(z - mu) for the entropy term
  float gz[N], gL[N * N], Ph[N * N], Yb[N * N], Xb[N * N], Qi[N * N];
  float red[8];
};

// sym <- 0.5 (X + X^T) + jitter I from global X
KV_DEV void load_sym_jitter(float *sym, const float *X, int n, float jitter) {
  KV_PAR(e, n * n) {
    const int i = e / n, j = e - i * n;
    sym[e] = 0.5f * (X[e] + X[j * n + i]) + (i == j ? jitter : 0.0f);
  }
  KV_SYNC();
}

// the diagonal fallback of _safe_cholesky (kalman_filter.py:298-302)
KV_DEV void chol_fallback(float *Lo, const float *X, int n) {
  KV_PAR(e, n * n) {
    const int i = e / n, j = e - i * n;
    Lo[e] = (i == j) ? sqrtf(fmaxf(X[e], 1e-6f)) : 0.0f;
  }
  KV_SYNC();
}

// factorise X (global, n x n) at the given whole-batch level into Lo; uses `sym` as scratch
KV_DEV void safe_chol_at_level(float *Lo, float *sym, const float *X, int n, int level) {
  if (level >= 5) {
    chol_fallback(Lo, X, n);
  } else {
    load_sym_jitter(sym, X, n, jitter_of_level(level));
    (void)cholesky(sym, Lo, n);
  }
}

KV_DEV int probe_level(float *Lo, float *sym, const float *X, int n) {
  for (int level = 0; level < 5; ++level) {
    load_sym_jitter(sym, X, n, jitter_of_level(level));
    if (cholesky(sym, Lo, n)) return level;
  }
  return 5;
}

#if defined(KVAE_HOSTSIM)
KV_DEV void atomic_max_i32(int32_t *p, int32_t v) { if (v > *p) *p = v; }
#else
// (a plain read first: when the whole batch needs a raised level, a hundred thousand wavefronts would otherwise queue on one
// address - 2.3 ms of atomics at the configs[4] shard - while all but the first few find their level already recorded)
KV_DEV void atomic_max_i32(int32_t *p, int32_t v) {
  if (*reinterpret_cast<volatile int32_t *>(p) >= v) return;
  atomicMax(p, v);
}
#endif

// ws (optional) receives per (b,t) the sample z_t = mu_t + L_t eps_t (n floats) of level 0: when the whole batch resolves to
// level 0 — the normal case — the main kernel reads z_{t-1}, z_{t+1} from it instead of re-factorising the two neighbouring
// steps (no re-reads of Sigma_s[t-1], Sigma_s[t+1]; its own factor it recomputes, which is cheaper than the 4 n^2 bytes each
// way that parking it would cost: HBM traffic of the pair of launches stays within ~1.1x of the algorithmic bytes).
template <class D>
KV_DEV void elbo_probe_body(const D d, const kvae_lgssm_problem &P, const float *Sig_s, const float *mus,
                            const float *eps, float *ws, int32_t *levels, int b, int t, ElboLds<D> &L) {
  const int n = d.n(), nn = n * n;
  const int64_t q0 = (int64_t)b * P.T + t;
  const int ls = probe_level(L.Ls, L.sym, Sig_s + q0 * nn, n);
  if (ws && ls == 0) {
    copy_in(L.eps, eps + q0 * n, n);
    copy_in(L.mu, mus + q0 * n, n);
    KV_SYNC();
    float *w = ws + q0 * n;
    KV_PAR(i, n) {
      float acc = L.mu[i];
      for (int k = 0; k <= i; ++k) acc = fmaf(L.Ls[i * n + k], L.eps[k], acc);
      w[i] = acc;
    }
    KV_SYNC();
  }
  int lq = 0;
  const bool q_shared = (P.Q.sb == 0 && P.Q.st == 0);
  if (t >= 1 && (!q_shared || (b == 0 && t == 1))) lq = probe_level(L.LQ, L.sym, stack_at(P.Q, b, t), n);
  KV_LANE0 {
    if (ls > 0) atomic_max_i32(levels + 0, ls);
    if (lq > 0) atomic_max_i32(levels + 1, lq);
  }
}

// v <- (L L^T)^{-1} x, returns |L^{-1} x|^2 through *quad (lane-0 serial, n <= 16); x,v in LDS.
KV_DEV void gauss_solve(const float *Lm, int n, const float *x, float *v, float *quad) {
  KV_LANE0 {
    float q = 0.f;
    for (int i = 0; i < n; ++i) {
      float acc = x[i];
      for (int k = 0; k < i; ++k) acc = fmaf(-Lm[i * n + k], v[k], acc);
      v[i] = acc / Lm[i * n + i];
      q = fmaf(v[i], v[i], q);
    }
    *quad = q;
    for (int i = n - 1; i >= 0; --i) {
      float acc = v[i];
      for (int k = i + 1; k < n; ++k) acc = fmaf(-Lm[k * n + i], v[k], acc);
      v[i] = acc / Lm[i * n + i];
    }
  }
}

KV_DEV float log_diag_sum(const float *Lm, int n) {
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += logf(Lm[i * n + i]);
  return s;
}

template <class D>
KV_DEV void elbo_body(const D d, const kvae_lgssm_problem &P, const float *mus, const float *Sigs,
                      const float *eps, float *terms, const int32_t *levels, const float *ws, float *g_mus,
                      float *g_Sigs, const kvae_lgssm_input_grads *Gp, int b, int t, ElboLds<D> &L) {
  const int n = d.n(), m = d.m(), p = d.p(), T = P.T, nn = n * n;
  const int64_t bT = (int64_t)b * T, q = bT + t;
  const int lvS = levels[0], lvQ = levels[1];
  const bool grads = (g_mus != nullptr);
  const bool has_prev = t >= 1, has_next = t + 1 < T;

  // ---- z_{t-1}, z_t, z_{t+1} = mu_s + chol(Sigma_s) eps  (kalman_filter.py:348-351) ----------
  const bool stashed = (ws != nullptr) && lvS == 0;   // the probe launch already sampled every step at level 0
  if (stashed) {
    safe_chol_at_level(L.Ls, L.sym, Sigs + q * nn, n, 0);
    for (int dt = -1; dt <= 1; ++dt) {
      const int tt = t + dt;
      if (tt >= 0 && tt < T) copy_in(L.z[dt + 1], ws + (bT + tt) * n, n);
    }
    KV_SYNC();
  }
  for (int dt = -1; dt <= 1 && !stashed; ++dt) {
    const int tt = t + dt;
    if (tt < 0 || tt >= T) continue;
    float *Lo = (dt == 0) ? L.Ls : L.Xb;  // only L_t is kept
    safe_chol_at_level(Lo, L.sym, Sigs + (bT + tt) * nn, n, lvS);
    copy_in(L.eps, eps + (bT + tt) * n, n);
    copy_in(L.mu, mus + (bT + tt) * n, n);
    KV_SYNC();
    KV_PAR(i, n) {
      float acc = L.mu[i];
      for (int k = 0; k <= i; ++k) acc = fmaf(Lo[i * n + k], L.eps[k], acc);
      L.z[dt + 1][i] = acc;
    }
    KV_SYNC();
  }
  // after the loop L.eps / L.mu hold step t+1 (or t when there is no next); reload step t
  copy_in(L.eps, eps + q * n, n);
  copy_in(L.mu, mus + q * n, n);
  // operands of step t (and t+1 for the gradient of z_t through the next transition)
  copy_in(L.C, stack_at(P.C, b, t), p * n);
  copy_in(L.y, P.Y + q * p, p);
  copy_in(L.R, P.R, p * p);
  if (has_prev) {
    copy_in(L.A, stack_at(P.A, b, t), nn);
    copy_in(L.Bm, stack_at(P.Bm, b, t), n * m);
    copy_in(L.u, P.U + q * m, m);
  }
  if (has_next && grads) {
    copy_in(L.An, stack_at(P.A, b, t + 1), nn);
    copy_in(L.Bn, stack_at(P.Bm, b, t + 1), n * m);
    copy_in(L.un, P.U + (q + 1) * m, m);
  }
  KV_SYNC();
  if (has_prev) safe_chol_at_level(L.LQ, L.sym, stack_at(P.Q, b, t), n, lvQ);
  if (has_next && grads) safe_chol_at_level(L.LQn, L.sym, stack_at(P.Q, b, t + 1), n, lvQ);
  // R and Sigma0 are factorised without jitter (MultivariateNormal(covariance_matrix=...), :373,:380)
  (void)cholesky(L.R, L.LR, p);
  if (t == 0) {
    copy_in(L.sym, P.Sigma0 + (int64_t)b * P.Sigma0_sb, nn);
    KV_SYNC();
    (void)cholesky(L.sym, L.L0, n);
  }
  // ---- residuals -----------------------------------------------------------------------------
  if (has_prev) {
    KV_PAR(i, n) {  // d_t = z_t - A_t z_{t-1} - B_t u_t  (:353-361)
      float acc = L.z[1][i];
      for (int k = 0; k < n; ++k) acc = fmaf(-L.A[i * n + k], L.z[0][k], acc);
      for (int k = 0; k < m; ++k) acc = fmaf(-L.Bm[i * m + k], L.u[k], acc);
      L.dt[i] = acc;
    }
  }
  if (has_next && grads) {
    KV_PAR(i, n) {
      float acc = L.z[2][i];
      for (int k = 0; k < n; ++k) acc = fmaf(-L.An[i * n + k], L.z[1][k], acc);
      for (int k = 0; k < m; ++k) acc = fmaf(-L.Bn[i * m + k], L.un[k], acc);
      L.dn[i] = acc;
    }
  }
  KV_PAR(i, p) {  // e_t = y_t - C_t z_t  (:372-374)
    float acc = L.y[i];
    for (int k = 0; k < n; ++k) acc = fmaf(-L.C[i * n + k], L.z[1][k], acc);
    L.em[i] = acc;
  }
  KV_PAR(i, n) {
    L.wv[i] = L.z[1][i] - L.mu[i];
    if (t == 0) L.di[i] = L.z[1][i] - P.mu0[(int64_t)b * P.mu0_sb + i];
  }
  KV_SYNC();
  // ---- quadratic forms (lane 0, serial) --------------------------------------------------------
  const float mv = *mask_addr(P, b, t);   // unconditional load from an always-valid address (see mask_addr)
  const float mk = P.mask ? mv : 1.0f;
  if (has_prev) gauss_solve(L.LQ, n, L.dt, L.vt, &L.red[0]);
  if (has_next && grads) gauss_solve(L.LQn, n, L.dn, L.vn, &L.red[1]);
  gauss_solve(L.LR, p, L.em, L.qe, &L.red[2]);
  if (t == 0) gauss_solve(L.L0, n, L.di, L.vi, &L.red[3]);
  gauss_solve(L.Ls, n, L.wv, L.gz /*scratch*/, &L.red[4]);
  KV_SYNC();
  KV_LANE0 {
    float tr = 0.f, em = 0.f, in = 0.f;
    if (has_prev) tr = -0.5f * (n * KV_LOG2PI + L.red[0]) - log_diag_sum(L.LQ, n);
    em = mk * (-0.5f * (p * KV_LOG2PI + L.red[2]) - log_diag_sum(L.LR, p));
    if (t == 0) in = -0.5f * (n * KV_LOG2PI + L.red[3]) - log_diag_sum(L.L0, n);
    const float ent = 0.5f * (n * KV_LOG2PI + L.red[4]) + log_diag_sum(L.Ls, n);
    terms[q * 4 + 0] = tr;
    terms[q * 4 + 1] = em;
    terms[q * 4 + 2] = in;
    terms[q * 4 + 3] = ent;
  }
  if (!grads) return;
  const kvae_lgssm_input_grads &G = *Gp;
  KV_SYNC();
  // ---- gradients of SUM(terms), unit upstream -------------------------------------------------
  // gz_t = -v_t [t>=1] + A_{t+1}^T v_{t+1} [t<T-1] + mask C_t^T R^{-1} e_t - Sigma0^{-1}(z_0-mu0) [t=0]
  // (the entropy's Mahalanobis part |L^{-1}(z-mu)|^2 = |eps|^2 has zero total derivative)
  KV_PAR(i, n) {
    float acc = 0.f;
    if (has_prev) acc -= L.vt[i];
    if (has_next) for (int k = 0; k < n; ++k) acc = fmaf(L.An[k * n + i], L.vn[k], acc);
    for (int k = 0; k < p; ++k) acc = fmaf(mk * L.C[k * n + i], L.qe[k], acc);
    if (t == 0) acc -= L.vi[i];
    L.gz[i] = acc;
  }
  KV_SYNC();
  KV_PAR(i, n) { g_mus[q * n + i] = L.gz[i]; }
  // input gradients that belong to step t
  KV_PAR(e, p * n) {  // gC_t = mask (R^{-1} e) z_t^T
    const int i = e / n, j = e - i * n;
    gstack_at(G.gC, b, t)[e] = mk * L.qe[i] * L.z[1][j];
  }
  KV_PAR(i, p) { G.gY[q * p + i] = -mk * L.qe[i]; }
  KV_PAR(e, nn) {  // gA_t = v_t z_{t-1}^T
    const int i = e / n, j = e - i * n;
    gstack_at(G.gA, b, t)[e] = has_prev ? L.vt[i] * L.z[0][j] : 0.0f;
  }
  KV_PAR(e, n * m) {
    const int i = e / m, j = e - i * m;
    gstack_at(G.gB, b, t)[e] = has_prev ? L.vt[i] * L.u[j] : 0.0f;
  }
  if (G.gU) {
    KV_PAR(i, m) {
      float acc = 0.f;
      if (has_prev) for (int k = 0; k < n; ++k) acc = fmaf(L.Bm[k * m + i], L.vt[k], acc);
      G.gU[q * m + i] = acc;
    }
  }
  if (G.gQ.ptr) {
    // gQ_t = -1/2 Q~^{-1} + 1/2 v v^T with Q~ = LQ LQ^T (symmetric, passes sym() unchanged);
    // diagonal-fallback level: only the un-clamped diagonal carries gradient.
    if (!has_prev) {
      KV_PAR(e, nn) { gstack_at(G.gQ, b, t)[e] = 0.0f; }
    } else if (lvQ >= 5) {
      const float *Qg = stack_at(P.Q, b, t);
      KV_PAR(e, nn) {
        const int i = e / n, j = e - i * n;
        float g = 0.f;
        if (i == j && Qg[e] >= 1e-6f) {
          const float l = L.LQ[e];                 // log N(d;0,diag l^2): d/dl = -1/l + d_i^2/l^3 ; dl/dq = 1/(2l)
          const float di = L.dt[i];
          g = (-1.0f / l + di * di / (l * l * l)) / (2.0f * l);
        }
        gstack_at(G.gQ, b, t)[e] = g;
      }
    } else {
      KV_PAR(e, nn) {  // Qi <- identity, then Qi <- LQ^{-1}
        const int i = e / n, j = e - i * n;
        L.Qi[e] = (i == j) ? 1.0f : 0.0f;
      }
      KV_SYNC();
      trisolve_lower(L.LQ, n, L.Qi, n, n);
      KV_SYNC();
      KV_PAR(e, nn) {
        const int i = e / n, j = e - i * n;
        float acc = 0.f;  // (LQ^{-T} LQ^{-1})[i,j] = sum_k Qi[k,i] Qi[k,j]
        for (int k = 0; k < n; ++k) acc = fmaf(L.Qi[k * n + i], L.Qi[k * n + j], acc);
        gstack_at(G.gQ, b, t)[e] = 0.5f * (L.vt[i] * L.vt[j] - acc);
      }
    }
  }
  // ---- through z = mu + L eps and the entropy's log-det into Sigma_s (Cholesky backward) ------
  if (lvS >= 5) {
    const float *Sg = Sigs + q * nn;
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float g = 0.f;
      if (i == j && Sg[e] >= 1e-6f) {
        const float l = L.Ls[e];
        g = (L.gz[i] * L.eps[i] + 1.0f / l) / (2.0f * l);
      }
      g_Sigs[q * nn + e] = g;
    }
    return;
  }
  // gL = tril(gz eps^T) + diag(1/L_ii) ;  Ph = Phi(L^T gL)
  KV_PAR(e, nn) {
    const int i = e / n, j = e - i * n;
    float g = 0.f;
    if (j <= i) g = L.gz[i] * L.eps[j];
    if (i == j) g += 1.0f / L.Ls[e];
    L.gL[e] = g;
  }
  KV_SYNC();
  KV_PAR(e, nn) {
    const int i = e / n, j = e - i * n;
    float acc = 0.f;
    if (j <= i) {
      for (int k = i; k < n; ++k) acc = fmaf(L.Ls[k * n + i], L.gL[k * n + j], acc);  // (L^T gL)[i,j], L lower
      if (i == j) acc *= 0.5f;
    }
    L.Ph[e] = acc;
  }
  KV_SYNC();
  trisolve_lower_t(L.Ls, n, L.Ph, n, n);  // Ph <- L^{-T} Phi
  KV_SYNC();
  KV_PAR(e, nn) {  // Yb = Ph^T so that the second solve is again column-wise
    const int i = e / n, j = e - i * n;
    L.Yb[e] = L.Ph[j * n + i];
  }
  KV_SYNC();
  trisolve_lower_t(L.Ls, n, L.Yb, n, n);  // Yb <- L^{-T} (L^{-T} Phi)^T = (L^{-T} Phi L^{-1})^T
  KV_SYNC();
  KV_PAR(e, nn) {
    const int i = e / n, j = e - i * n;
    g_Sigs[q * nn + e] = 0.5f * (L.Yb[e] + L.Yb[j * n + i]);
  }
}

}  // namespace kvae
