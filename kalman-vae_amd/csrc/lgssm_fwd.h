// lgssm_fwd.h — forward sweeps of the LGSSM for ONE sequence handled by ONE wavefront:
//   filter_sweep : Kalman filter, t = 0..T-1      (reference: kalman_filter.py:31-104,151-185)
//   rts_sweep    : RTS smoother,  t = T-2..0      (reference: kalman_filter.py:204-237,249-272)
// State and every intermediate tile live in the per-wave LDS scratch FwdLds; the per-step
// operands (A_t,B_t,C_t,Q_t,y_t,u_t,mask_t) are prefetched one step ahead into registers.
#pragma once
#include "lgssm_vm.h"

namespace kvae {

template <class D>
struct FwdLds {
  static constexpr int N = D::NMAX, M = D::MMAX, P = D::PMAX;
  // per-step operands
  float A[N * N], Bm[N * M], C[P * N], Q[N * N], R[P * P], y[P], u[M], mk[1];
  // carried belief and prediction
  float mu[N], Sig[N * N], mup[N], Sigp[N * N];
  // filter intermediates
  float AS[N * N], r[P], CP[P * N], PCT[N * P], aug[P * (P + N)], Kt[P * N];
  float IKC[N * N], KR[N * P], T1[N * N], F0[N * N];
  // smoother intermediates
  float mus[N], Sigs[N * N], aug2[N * 2 * N], Xs[N * N], muf[N], dmu[N], Dm[N * N], TJ[N * N], M2[N * N];
};

template <class D>
struct StepOperands {  // register prefetch of step t+1 while step t computes (static dims only)
  Prefetch<D::NMAX * D::NMAX> a, q;
  Prefetch<D::NMAX * D::MMAX> b;
  Prefetch<D::PMAX * D::NMAX> c;
  Prefetch<D::PMAX> y;
  Prefetch<D::MMAX> u;
  Prefetch<1> mk;
};

KV_DEV const float *stack_at(const kvae_stack &s, int b, int t) { return s.ptr + (int64_t)b * s.sb + (int64_t)t * s.st; }
KV_DEV float *gstack_at(const kvae_gstack &s, int b, int t) { return s.ptr + (int64_t)b * s.sb + (int64_t)t * s.st; }

// Address of mask[b,t], or of a harmless valid float (R[0]) when every frame is observed (mask == NULL).
// The single-element mask load is wave-uniform, so hipcc turns it into a SCALAR load; scalar loads
// ignore EXEC and hipcc may drop the skip-branch around a short predicated block (observed on
// ROCm 7.2: `if (mask) v = mask[i]` under `lane == 0` faulted on address nil).  Hence: always load
// from a valid address, select afterwards.
KV_DEV const float *mask_addr(const kvae_lgssm_problem &P, int b, int t) {
  const int tc = t < P.T ? t : P.T - 1;  // a dropped skip-branch must not read past the last step either
  return P.mask ? P.mask + (int64_t)b * P.T + tc : P.R;
}

template <class D>
KV_DEV void operands_issue(const D d, const kvae_lgssm_problem &P, int b, int t, StepOperands<D> &pf) {
  const int n = d.n(), m = d.m(), p = d.p();
  pf.a.issue(stack_at(P.A, b, t), n * n);
  pf.b.issue(stack_at(P.Bm, b, t), n * m);
  pf.c.issue(stack_at(P.C, b, t), p * n);
  pf.q.issue(stack_at(P.Q, b, t), n * n);
  pf.y.issue(P.Y + ((int64_t)b * P.T + t) * p, p);
  pf.u.issue(P.U + ((int64_t)b * P.T + t) * m, m);
  pf.mk.issue(mask_addr(P, b, t), 1);
}

template <class D, class LDS>
KV_DEV void operands_commit(const D d, const kvae_lgssm_problem &P, const StepOperands<D> &pf, LDS &L) {
  const int n = d.n(), m = d.m(), p = d.p();
  pf.a.commit(L.A, n * n);
  pf.b.commit(L.Bm, n * m);
  pf.c.commit(L.C, p * n);
  pf.q.commit(L.Q, n * n);
  pf.y.commit(L.y, p);
  pf.u.commit(L.u, m);
  KV_LANE0 { L.mk[0] = P.mask ? pf.mk.v[0] : 1.0f; }
}

template <class D, class LDS>
KV_DEV void operands_load(const D d, const kvae_lgssm_problem &P, int b, int t, LDS &L) {
  const int n = d.n(), m = d.m(), p = d.p();
  copy_in(L.A, stack_at(P.A, b, t), n * n);
  copy_in(L.Bm, stack_at(P.Bm, b, t), n * m);
  copy_in(L.C, stack_at(P.C, b, t), p * n);
  copy_in(L.Q, stack_at(P.Q, b, t), n * n);
  copy_in(L.y, P.Y + ((int64_t)b * P.T + t) * p, p);
  copy_in(L.u, P.U + ((int64_t)b * P.T + t) * m, m);
  const float mv = *mask_addr(P, b, t);
  KV_LANE0 { L.mk[0] = P.mask ? mv : 1.0f; }
}

// Recompute-able part of one filter step, shared by the forward and the backward sweep:
// from (mu,Sig) = belief at t-1 and the operands in LDS, produce mup,Sigp (if PREDICT), r, CP,
// PCT, S (returned in Ssave when non-null), Kt = S^{-1} PCT^T (unmasked, [p,n]).
template <class D, class LDS>
KV_DEV void filter_gain(const D d, LDS &L, bool predict, float *Ssave) {
  const int n = d.n(), m = d.m(), p = d.p();
  // phase 1: mup = A mu + B u ; AS = A Sig
  if (predict) {
    KV_PAR(i, n) {
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.A[i * n + k], L.mu[k], acc);
      for (int k = 0; k < m; ++k) acc = fmaf(L.Bm[i * m + k], L.u[k], acc);
      L.mup[i] = acc;
    }
  }
  KV_PAR(e, n * n) {
    const int i = e / n, j = e - i * n;
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc = fmaf(L.A[i * n + k], L.Sig[k * n + j], acc);
    L.AS[e] = acc;
  }
  KV_SYNC();
  // phase 2: Sigp = AS A^T + Q ; r = y - C mup
  if (predict) {
    KV_PAR(e, n * n) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.AS[i * n + k], L.A[j * n + k], acc);
      L.Sigp[e] = acc + L.Q[e];
    }
  }
  KV_PAR(i, p) {
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc = fmaf(L.C[i * n + k], L.mup[k], acc);
    L.r[i] = L.y[i] - acc;
  }
  KV_SYNC();
  // phase 3: CP = C Sigp ; PCT = Sigp C^T
  KV_PAR(e, p * n) {
    const int i = e / n, j = e - i * n;
    float acc = 0.f, acc2 = 0.f;
    for (int k = 0; k < n; ++k) {
      acc = fmaf(L.C[i * n + k], L.Sigp[k * n + j], acc);
      acc2 = fmaf(L.Sigp[j * n + k], L.C[i * n + k], acc2);
    }
    L.CP[e] = acc;           // CP[i,j]
    L.PCT[j * p + i] = acc2; // PCT[j,i]
  }
  KV_SYNC();
  // phase 4: aug = [ sym(CP C^T + R) | PCT^T ]
  const int ld = p + n;
  KV_PAR(e, p * p) {
    const int i = e / p, j = e - i * p;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < n; ++k) {
      s1 = fmaf(L.CP[i * n + k], L.C[j * n + k], s1);
      s2 = fmaf(L.CP[j * n + k], L.C[i * n + k], s2);
    }
    const float s = 0.5f * ((s1 + L.R[i * p + j]) + (s2 + L.R[j * p + i]));
    L.aug[i * ld + j] = s;
    if (Ssave) Ssave[e] = s;
  }
  KV_PAR(e, p * n) {
    const int i = e / n, j = e - i * n;
    L.aug[i * ld + p + j] = L.PCT[j * p + i];
  }
  KV_SYNC();
  lu_solve(L.aug, p, n, L.Kt, n);  // Kt[p,n]
}

// One predict + update on the operands and belief held in LDS (the body of the filter loop, shared with the
// masked alpha-network kernel in kvae_lgssm_wide.hip): writes pred/filt outputs of step bt and carries (mu, Sig).
template <class D, class LDS>
KV_DEV void filter_step_core(const D d, const kvae_lgssm_states &S, int64_t bt, LDS &L) {
  const int n = d.n(), p = d.p(), nn = n * n;
  filter_gain(d, L, true, nullptr);
  const float mk = L.mk[0];
  // predicted belief is final here: stream it out
  copy_out(S.mus_pred + bt * n, L.mup, n);
  copy_out(S.Sigmas_pred + bt * nn, L.Sigp, nn);
  // phase: K = mask * Kt^T ; IKC = I - K C ; KR = K R ; mu_f = mup + K r
  KV_PAR(e, nn) {
    const int i = e / n, j = e - i * n;
    float acc = 0.f;
    for (int k = 0; k < p; ++k) acc = fmaf(mk * L.Kt[k * n + i], L.C[k * n + j], acc);
    L.IKC[e] = (i == j ? 1.0f : 0.0f) - acc;
  }
  KV_PAR(e, n * p) {
    const int i = e / p, j = e - i * p;
    float acc = 0.f;
    for (int k = 0; k < p; ++k) acc = fmaf(mk * L.Kt[k * n + i], L.R[k * p + j], acc);
    L.KR[e] = acc;
  }
  KV_PAR(i, n) {
    float acc = L.mup[i];
    for (int k = 0; k < p; ++k) acc = fmaf(mk * L.Kt[k * n + i], L.r[k], acc);
    L.muf[i] = acc;
  }
  KV_SYNC();
  // phase: T1 = IKC Sigp
  KV_PAR(e, nn) {
    const int i = e / n, j = e - i * n;
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc = fmaf(L.IKC[i * n + k], L.Sigp[k * n + j], acc);
    L.T1[e] = acc;
  }
  KV_SYNC();
  // phase: F0 = T1 IKC^T + KR K^T (Joseph form)
  KV_PAR(e, nn) {
    const int i = e / n, j = e - i * n;
    float acc = 0.f;
    for (int k = 0; k < n; ++k) acc = fmaf(L.T1[i * n + k], L.IKC[j * n + k], acc);
    float acc2 = 0.f;
    for (int k = 0; k < p; ++k) acc2 = fmaf(L.KR[i * p + k], mk * L.Kt[k * n + j], acc2);
    L.F0[e] = acc + acc2;
  }
  KV_SYNC();
  // phase: Sig_f = sym(F0); carry (mu_f, Sig_f) to the next step and stream them out
  KV_PAR(e, nn) {
    const int i = e / n, j = e - i * n;
    const float v = 0.5f * (L.F0[e] + L.F0[j * n + i]);
    L.Sig[e] = v;
    S.Sigmas_filt[bt * nn + e] = v;
  }
  KV_PAR(i, n) {
    const float v = L.muf[i];
    L.mu[i] = v;
    S.mus_filt[bt * n + i] = v;
  }
  KV_SYNC();
}

template <class D>
KV_DEV void filter_sweep(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, FwdLds<D> &L) {
  const int n = d.n(), m = d.m(), p = d.p(), T = P.T, nn = n * n;
  (void)m;
  copy_in(L.mu, P.mu0 + (int64_t)b * P.mu0_sb, n);
  copy_in(L.Sig, P.Sigma0 + (int64_t)b * P.Sigma0_sb, nn);
  copy_in(L.R, P.R, p * p);
  KV_LANE0 { L.mk[0] = 1.0f; }
  StepOperands<D> pf;
  if (D::is_static) operands_issue(d, P, b, 0, pf);
  for (int t = 0; t < T; ++t) {
    if (D::is_static) {
      operands_commit(d, P, pf, L);
      if (t + 1 < T) operands_issue(d, P, b, t + 1, pf);
    } else {
      operands_load(d, P, b, t, L);
    }
    KV_SYNC();
    filter_step_core(d, S, (int64_t)b * T + t, L);
  }
}

// RTS sweep.  Reads the filtered/predicted stacks of sequence b back from global memory (they
// were just written by this same wavefront, or by an earlier launch) and writes the smoothed ones.
template <class D>
KV_DEV void rts_sweep(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, FwdLds<D> &L) {
  const int n = d.n(), T = P.T, nn = n * n;
  const int64_t bT = (int64_t)b * T;
  // last step: smoothed = filtered (kalman_filter.py:251-256)
  copy_in(L.mus, S.mus_filt + (bT + T - 1) * n, n);
  copy_in(L.Sigs, S.Sigmas_filt + (bT + T - 1) * nn, nn);
  KV_SYNC();
  copy_out(S.mus_smooth + (bT + T - 1) * n, L.mus, n);
  copy_out(S.Sigmas_smooth + (bT + T - 1) * nn, L.Sigs, nn);
  Prefetch<D::NMAX * D::NMAX> pfS, pfP, pfA;
  Prefetch<D::NMAX> pfm, pfq;
  if (D::is_static && T >= 2) {
    pfS.issue(S.Sigmas_filt + (bT + T - 2) * nn, nn);
    pfP.issue(S.Sigmas_pred + (bT + T - 1) * nn, nn);
    pfA.issue(stack_at(P.A, b, T - 1), nn);
    pfm.issue(S.mus_filt + (bT + T - 2) * n, n);
    pfq.issue(S.mus_pred + (bT + T - 1) * n, n);
  }
  for (int t = T - 2; t >= 0; --t) {
    // operands: Sig_f[t] -> Sig, Sig_p[t+1] -> Sigp, A[t+1] -> A, mu_f[t] -> muf, mu_p[t+1] -> mup
    if (D::is_static) {
      pfS.commit(L.Sig, nn);
      pfP.commit(L.Sigp, nn);
      pfA.commit(L.A, nn);
      pfm.commit(L.muf, n);
      pfq.commit(L.mup, n);
      if (t >= 1) {
        pfS.issue(S.Sigmas_filt + (bT + t - 1) * nn, nn);
        pfP.issue(S.Sigmas_pred + (bT + t) * nn, nn);
        pfA.issue(stack_at(P.A, b, t), nn);
        pfm.issue(S.mus_filt + (bT + t - 1) * n, n);
        pfq.issue(S.mus_pred + (bT + t) * n, n);
      }
    } else {
      copy_in(L.Sig, S.Sigmas_filt + (bT + t) * nn, nn);
      copy_in(L.Sigp, S.Sigmas_pred + (bT + t + 1) * nn, nn);
      copy_in(L.A, stack_at(P.A, b, t + 1), nn);
      copy_in(L.muf, S.mus_filt + (bT + t) * n, n);
      copy_in(L.mup, S.mus_pred + (bT + t + 1) * n, n);
    }
    KV_SYNC();
    // phase: aug2 = [ Sigp^T | W^T ], W = Sig_f A^T ; dmu ; D
    const int ld = 2 * n;
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      L.aug2[i * ld + j] = L.Sigp[j * n + i];
      float acc = 0.f;  // W^T[i,j] = W[j,i] = sum_k Sig_f[j,k] A[i,k]
      for (int k = 0; k < n; ++k) acc = fmaf(L.Sig[j * n + k], L.A[i * n + k], acc);
      L.aug2[i * ld + n + j] = acc;
      L.Dm[e] = L.Sigs[e] - L.Sigp[e];
    }
    KV_PAR(i, n) { L.dmu[i] = L.mus[i] - L.mup[i]; }
    KV_SYNC();
    lu_solve(L.aug2, n, n, L.Xs, n);  // Xs = J^T
    // phase: TJ = J D ; mu_s = mu_f + J dmu
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.Xs[k * n + i], L.Dm[k * n + j], acc);
      L.TJ[e] = acc;
    }
    KV_PAR(i, n) {
      float acc = L.muf[i];
      for (int k = 0; k < n; ++k) acc = fmaf(L.Xs[k * n + i], L.dmu[k], acc);
      L.mu[i] = acc;  // staging: mus is still being read in this phase by nobody, but keep it clean
    }
    KV_SYNC();
    // phase: M2 = Sig_f + TJ J^T
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.TJ[i * n + k], L.Xs[k * n + j], acc);
      L.M2[e] = L.Sig[e] + acc;
    }
    KV_SYNC();
    // phase: Sig_s = sym(M2); carry and stream out
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      const float v = 0.5f * (L.M2[e] + L.M2[j * n + i]);
      L.Sigs[e] = v;
      S.Sigmas_smooth[(bT + t) * nn + e] = v;
    }
    KV_PAR(i, n) {
      const float v = L.mu[i];
      L.mus[i] = v;
      S.mus_smooth[(bT + t) * n + i] = v;
    }
    KV_SYNC();
  }
}

}  // namespace kvae
