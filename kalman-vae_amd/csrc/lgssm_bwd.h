// lgssm_bwd.h — hand-derived reverse mode of the filter + RTS smoother for ONE sequence handled by
// ONE wavefront.  The reference obtains these gradients from autograd over ~580 aten calls per
// time step (kalman_filter.py:151-185, 257-272); here they are two in-kernel sweeps:
//
//   rts_bwd_sweep    t = 0..T-2   adjoint of the (reverse-time) smoother loop; hands the adjoints of
//                                 (mu_f, Sig_f, mu_p, Sig_p) to the filter sweep through `ws`
//                                 and leaves the smoother's share of gA in the gA output
//   filter_bwd_sweep t = T-1..0   adjoint of the filter loop; produces gA,gB,gC,gQ,gY,gU (+ g of
//                                 the initial belief)
//
// Forward intermediates (gain, S, J) are recomputed from the saved beliefs, not stored.
//
// ws layout per (b,t): [ g_mu_f (n) | g_Sig_f (n*n) | g_mu_p (n) | g_Sig_p (n*n) ].
#pragma once
#include "lgssm_fwd.h"

namespace kvae {

template <class D>
struct BwdLds {
  static constexpr int N = D::NMAX, M = D::MMAX, P = D::PMAX;
  // operands + forward recomputation (same names as FwdLds so filter_gain() can be shared)
  float A[N * N], Bm[N * M], C[P * N], Q[N * N], R[P * P], y[P], u[M], mk[1];
  float mu[N], Sig[N * N], mup[N], Sigp[N * N];
  float AS[N * N], r[P], CP[P * N], PCT[N * P], aug[P * (P + N)], Kt[P * N];
  float IKC[N * N], Ssv[P * P];
  // filter adjoints
  float gmu[N], gSig[N * N];                 // carried adjoint of the filtered belief at t
  float G[N * N], X1[N * N], GK[N * P], gr[P], gIKC[N * N], gSp[N * N], gK[N * P], gC[P * N];
  float Z[P * N], gS0[P * P], gCP[P * N], gmp[N], gAS[N * N];
  // smoother part
  float mus[N], Sigs[N * N], aug2[N * 2 * N], Xs[N * N], dmu[N], Dm[N * N];
  float gsm[N], gsS[N * N];                  // carried adjoint of the smoothed belief at t
  float gM[N * N], Y1[N * N], gJ[N * N], gD[N * N], gdm[N], gR[N * N];
};

KV_DEV float opt_load(const float *p, int64_t idx) { return p ? p[idx] : 0.0f; }

template <class D>
KV_DEV void rts_bwd_sweep(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                          const kvae_lgssm_states &U, const kvae_lgssm_input_grads &G, float *ws, int b,
                          BwdLds<D> &L) {
  const int n = d.n(), T = P.T, nn = n * n, rec = 2 * (n + nn);
  const int64_t bT = (int64_t)b * T;
  float *w = ws + bT * rec;
  // adjoint of the smoothed belief at t = 0 is just the upstream gradient
  KV_PAR(i, n) { L.gsm[i] = opt_load(U.mus_smooth, bT * n + i); }
  KV_PAR(e, nn) { L.gsS[e] = opt_load(U.Sigmas_smooth, bT * nn + e); }
  // predicted belief at t = 0 only has upstream adjoints
  KV_PAR(i, n) { w[n + nn + i] = opt_load(U.mus_pred, bT * n + i); }
  KV_PAR(e, nn) { w[n + nn + n + e] = opt_load(U.Sigmas_pred, bT * nn + e); }
  KV_PAR(e, nn) { gstack_at(G.gA, b, 0)[e] = 0.0f; }
  KV_SYNC();
  for (int t = 0; t + 1 < T; ++t) {
    const int64_t q = bT + t;
    copy_in(L.Sig, S.Sigmas_filt + q * nn, nn);
    copy_in(L.Sigp, S.Sigmas_pred + (q + 1) * nn, nn);
    copy_in(L.A, stack_at(P.A, b, t + 1), nn);
    copy_in(L.mup, S.mus_pred + (q + 1) * n, n);
    copy_in(L.mus, S.mus_smooth + (q + 1) * n, n);
    copy_in(L.Sigs, S.Sigmas_smooth + (q + 1) * nn, nn);
    KV_SYNC();
    // recompute J: aug2 = [Sigp^T | (Sig_f A^T)^T]; dmu; D; gM = sym(gsS)
    const int ld = 2 * n;
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      L.aug2[i * ld + j] = L.Sigp[j * n + i];
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.Sig[j * n + k], L.A[i * n + k], acc);
      L.aug2[i * ld + n + j] = acc;
      L.Dm[e] = L.Sigs[e] - L.Sigp[e];
      L.gM[e] = 0.5f * (L.gsS[e] + L.gsS[j * n + i]);
    }
    KV_PAR(i, n) { L.dmu[i] = L.mus[i] - L.mup[i]; }
    KV_SYNC();
    lu_solve(L.aug2, n, n, L.Xs, n);  // Xs = J^T, i.e. J[i,k] = Xs[k*n+i]
    // Y1 = gM J ; gdm = J^T gsm
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.gM[i * n + k], L.Xs[j * n + k], acc);
      L.Y1[e] = acc;
    }
    KV_PAR(i, n) {
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.Xs[i * n + k], L.gsm[k], acc);
      L.gdm[i] = acc;
    }
    KV_SYNC();
    // gJ = Y1 (D^T + D) + gsm dmu^T  -> aug2 = [Sigp | gJ^T] ;  gD = J^T Y1
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.Y1[i * n + k], L.Dm[j * n + k] + L.Dm[k * n + j], acc);
      acc = fmaf(L.gsm[i], L.dmu[j], acc);
      L.gJ[e] = acc;                       // gJ[i,j]
      L.aug2[j * ld + n + i] = acc;        // gJ^T
      L.aug2[i * ld + j] = L.Sigp[e];
      float a2 = 0.f;                      // gD[i,j] = sum_k J[k,i] Y1[k,j]
      for (int k = 0; k < n; ++k) a2 = fmaf(L.Xs[i * n + k], L.Y1[k * n + j], a2);
      L.gD[e] = a2;
    }
    KV_SYNC();
    lu_solve(L.aug2, n, n, L.gR, n);  // gR = Sigp^{-1} gJ^T = gW^T
    // hand-offs to the filter sweep and to the next smoother step
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      // g_Sig_f[t] = up + gM + gW A,  gW A [i,j] = sum_k gR[k,i] A[k,j]
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.gR[k * n + i], L.A[k * n + j], acc);
      w[n + e] = opt_load(U.Sigmas_filt, q * nn + e) + L.gM[e] + acc;
      // gP[i,j] = -sum_k J[k,i] gW[k,j] = -sum_k Xs[i*n+k] gR[j*n+k]
      float gp = 0.f;
      for (int k = 0; k < n; ++k) gp = fmaf(L.Xs[i * n + k], L.gR[j * n + k], gp);
      w[rec + n + nn + n + e] = opt_load(U.Sigmas_pred, (q + 1) * nn + e) - L.gD[e] - gp;
      // smoother share of gA[t+1] = gW^T Sig_f = gR Sig_f
      float ga = 0.f;
      for (int k = 0; k < n; ++k) ga = fmaf(L.gR[i * n + k], L.Sig[k * n + j], ga);
      gstack_at(G.gA, b, t + 1)[e] = ga;
    }
    KV_PAR(i, n) {
      w[i] = opt_load(U.mus_filt, q * n + i) + L.gsm[i];
      w[rec + n + nn + i] = opt_load(U.mus_pred, (q + 1) * n + i) - L.gdm[i];
    }
    KV_SYNC();
    // carried adjoint of the smoothed belief at t+1
    KV_PAR(e, nn) { L.gsS[e] = opt_load(U.Sigmas_smooth, (q + 1) * nn + e) + L.gD[e]; }
    KV_PAR(i, n) { L.gsm[i] = opt_load(U.mus_smooth, (q + 1) * n + i) + L.gdm[i]; }
    KV_SYNC();
    w += rec;
  }
  // t = T-1: smoothed == filtered
  const int64_t q = bT + T - 1;
  KV_PAR(e, nn) { w[n + e] = opt_load(U.Sigmas_filt, q * nn + e) + L.gsS[e]; }
  KV_PAR(i, n) { w[i] = opt_load(U.mus_filt, q * n + i) + L.gsm[i]; }
  KV_SYNC();
}

// Filter-only variant of the hand-off: ws <- upstream adjoints of the filt/pred stacks.
template <class D>
KV_DEV void filter_bwd_seed(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &U,
                            const kvae_lgssm_input_grads &G, float *ws, int b) {
  const int n = d.n(), T = P.T, nn = n * n, rec = 2 * (n + nn);
  const int64_t bT = (int64_t)b * T;
  for (int t = 0; t < T; ++t) {
    float *w = ws + (bT + t) * rec;
    KV_PAR(i, n) {
      w[i] = opt_load(U.mus_filt, (bT + t) * n + i);
      w[n + nn + i] = opt_load(U.mus_pred, (bT + t) * n + i);
    }
    KV_PAR(e, nn) {
      w[n + e] = opt_load(U.Sigmas_filt, (bT + t) * nn + e);
      w[n + nn + n + e] = opt_load(U.Sigmas_pred, (bT + t) * nn + e);
      gstack_at(G.gA, b, t)[e] = 0.0f;
    }
  }
  KV_SYNC();
}

// The filter adjoint as three pieces, so that a caller can interleave other work between time steps (the fused
// alpha-network backward of kvae_lgssm_wide.hip): begin -> step(T-1) ... step(0) -> end.
template <class D>
KV_DEV void filter_bwd_begin(const D d, const kvae_lgssm_problem &P, BwdLds<D> &L) {
  const int n = d.n(), p = d.p(), nn = n * n;
  copy_in(L.R, P.R, p * p);
  KV_LANE0 { L.mk[0] = 1.0f; }
  KV_PAR(i, n) { L.gmu[i] = 0.0f; }
  KV_PAR(e, nn) { L.gSig[e] = 0.0f; }
  KV_SYNC();
}

// One step of the filter adjoint: consumes the carried (gmu, gSig) of step t+1 and the hand-off record ws[b,t], writes
// gA/gB/gC/gQ/gY/gU of step t and leaves (gmu, gSig) for step t-1.  On return L.mup / L.C / L.mk hold step t's values.
template <class D>
KV_DEV void filter_bwd_step(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                            const kvae_lgssm_input_grads &G, const float *ws, int b, int t, BwdLds<D> &L) {
  const int n = d.n(), m = d.m(), p = d.p(), T = P.T, nn = n * n, rec = 2 * (n + nn);
  const int64_t bT = (int64_t)b * T;
  {
    const int64_t q = bT + t;
    const float *w = ws + q * rec;
    operands_load(d, P, b, t, L);
    if (t > 0) {
      copy_in(L.mu, S.mus_filt + (q - 1) * n, n);
      copy_in(L.Sig, S.Sigmas_filt + (q - 1) * nn, nn);
    } else {
      copy_in(L.mu, P.mu0 + (int64_t)b * P.mu0_sb, n);
      copy_in(L.Sig, P.Sigma0 + (int64_t)b * P.Sigma0_sb, nn);
    }
    copy_in(L.mup, S.mus_pred + q * n, n);
    copy_in(L.Sigp, S.Sigmas_pred + q * nn, nn);
    // total adjoint of the filtered belief at t = carried (from step t+1) + handed-off
    KV_PAR(i, n) { L.gmu[i] += w[i]; }
    KV_PAR(e, nn) { L.gSig[e] += w[n + e]; }
    KV_SYNC();
    filter_gain(d, L, false, L.Ssv);  // AS, r, CP, PCT, S (-> Ssv), Kt
    const float mk = L.mk[0];
    // IKC = I - K C ; G = sym(gSig) ; gr = K^T gmu
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < p; ++k) acc = fmaf(mk * L.Kt[k * n + i], L.C[k * n + j], acc);
      L.IKC[e] = (i == j ? 1.0f : 0.0f) - acc;
      L.G[e] = 0.5f * (L.gSig[e] + L.gSig[j * n + i]);
    }
    KV_PAR(i, p) {
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(mk * L.Kt[i * n + k], L.gmu[k], acc);
      L.gr[i] = acc;
    }
    KV_SYNC();
    // X1 = G IKC ; GK = G K
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.G[i * n + k], L.IKC[k * n + j], acc);
      L.X1[e] = acc;
    }
    KV_PAR(e, n * p) {
      const int i = e / p, j = e - i * p;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.G[i * n + k], mk * L.Kt[j * n + k], acc);
      L.GK[e] = acc;
    }
    KV_SYNC();
    // gIKC = X1 (Sigp^T + Sigp) ; gSp = IKC^T X1 + handed-off adjoint of Sig_p
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f, a2 = 0.f;
      for (int k = 0; k < n; ++k) {
        acc = fmaf(L.X1[i * n + k], L.Sigp[j * n + k] + L.Sigp[k * n + j], acc);
        a2 = fmaf(L.IKC[k * n + i], L.X1[k * n + j], a2);
      }
      L.gIKC[e] = acc;
      L.gSp[e] = a2 + w[n + nn + n + e];
    }
    KV_SYNC();
    // gK = GK (R^T + R) - gIKC C^T + gmu r^T ; gC = -K^T gIKC
    KV_PAR(e, n * p) {
      const int i = e / p, j = e - i * p;
      float acc = 0.f;
      for (int k = 0; k < p; ++k) acc = fmaf(L.GK[i * p + k], L.R[j * p + k] + L.R[k * p + j], acc);
      for (int k = 0; k < n; ++k) acc = fmaf(-L.gIKC[i * n + k], L.C[j * n + k], acc);
      acc = fmaf(L.gmu[i], L.r[j], acc);
      L.gK[e] = acc;
    }
    KV_PAR(e, p * n) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(mk * L.Kt[i * n + k], L.gIKC[k * n + j], acc);
      L.gC[e] = -acc;
    }
    KV_SYNC();
    // Z = solve(S^T, mask * gK^T)   [p,n]
    const int ld = p + n;
    KV_PAR(e, p * p) {
      const int i = e / p, j = e - i * p;
      L.aug[i * ld + j] = L.Ssv[j * p + i];
    }
    KV_PAR(e, p * n) {
      const int i = e / n, j = e - i * n;
      L.aug[i * ld + p + j] = mk * L.gK[j * p + i];
    }
    KV_SYNC();
    lu_solve(L.aug, p, n, L.Z, n);
    // gS0 = sym(-Z Kt^T)
    KV_PAR(e, p * p) {
      const int i = e / p, j = e - i * p;
      float s1 = 0.f, s2 = 0.f;
      for (int k = 0; k < n; ++k) {
        s1 = fmaf(L.Z[i * n + k], L.Kt[j * n + k], s1);
        s2 = fmaf(L.Z[j * n + k], L.Kt[i * n + k], s2);
      }
      L.gS0[e] = -0.5f * (s1 + s2);
    }
    KV_SYNC();
    // gCP = gS0 C
    KV_PAR(e, p * n) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < p; ++k) acc = fmaf(L.gS0[i * p + k], L.C[k * n + j], acc);
      L.gCP[e] = acc;
    }
    KV_SYNC();
    // gSp += Z^T C + C^T gCP ; gC += Z Sigp + gS0 CP + gCP Sigp^T - gr mup^T ; gmp ; gY
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = L.gSp[e];
      for (int k = 0; k < p; ++k) {
        acc = fmaf(L.Z[k * n + i], L.C[k * n + j], acc);
        acc = fmaf(L.C[k * n + i], L.gCP[k * n + j], acc);
      }
      L.gSp[e] = acc;  // same element read and written by the same lane only
    }
    KV_PAR(e, p * n) {
      const int i = e / n, j = e - i * n;
      float acc = L.gC[e];
      for (int k = 0; k < n; ++k) {
        acc = fmaf(L.Z[i * n + k], L.Sigp[k * n + j], acc);
        acc = fmaf(L.gCP[i * n + k], L.Sigp[j * n + k], acc);
      }
      for (int k = 0; k < p; ++k) acc = fmaf(L.gS0[i * p + k], L.CP[k * n + j], acc);
      acc = fmaf(-L.gr[i], L.mup[j], acc);
      gstack_at(G.gC, b, t)[e] = acc;
    }
    KV_PAR(i, n) {
      float acc = L.gmu[i] + w[n + nn + i];
      for (int k = 0; k < p; ++k) acc = fmaf(-L.C[k * n + i], L.gr[k], acc);
      L.gmp[i] = acc;
    }
    KV_PAR(i, p) { G.gY[q * p + i] = L.gr[i]; }
    KV_SYNC();
    // gAS = gSp A ; gQ = gSp
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.gSp[i * n + k], L.A[k * n + j], acc);
      L.gAS[e] = acc;
      if (G.gQ.ptr) gstack_at(G.gQ, b, t)[e] = L.gSp[e];
    }
    KV_SYNC();
    // gA += gSp^T AS + gAS Sig^T + gmp mu^T ; carried adjoints for t-1 ; gB ; gU
    KV_PAR(e, nn) {
      const int i = e / n, j = e - i * n;
      float acc = gstack_at(G.gA, b, t)[e];
      for (int k = 0; k < n; ++k) {
        acc = fmaf(L.gSp[k * n + i], L.AS[k * n + j], acc);
        acc = fmaf(L.gAS[i * n + k], L.Sig[j * n + k], acc);
      }
      acc = fmaf(L.gmp[i], L.mu[j], acc);
      gstack_at(G.gA, b, t)[e] = acc;
      float gs = 0.f;  // (A^T gAS)[i,j]
      for (int k = 0; k < n; ++k) gs = fmaf(L.A[k * n + i], L.gAS[k * n + j], gs);
      L.gSig[e] = gs;
    }
    KV_PAR(i, n) {
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.A[k * n + i], L.gmp[k], acc);
      L.gmu[i] = acc;
    }
    KV_PAR(e, n * m) {
      const int i = e / m, j = e - i * m;
      gstack_at(G.gB, b, t)[e] = L.gmp[i] * L.u[j];
    }
    if (G.gU) {
      KV_PAR(i, m) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(L.Bm[k * m + i], L.gmp[k], acc);
        G.gU[q * m + i] = acc;
      }
    }
    KV_SYNC();
  }
}

template <class D>
KV_DEV void filter_bwd_end(const D d, const kvae_lgssm_input_grads &G, int b, BwdLds<D> &L) {
  const int n = d.n(), nn = n * n;
  if (G.g_mu0) copy_out(G.g_mu0 + (int64_t)b * n, L.gmu, n);
  if (G.g_Sigma0) copy_out(G.g_Sigma0 + (int64_t)b * nn, L.gSig, nn);
}

template <class D>
KV_DEV void filter_bwd_sweep(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                             const kvae_lgssm_input_grads &G, const float *ws, int b, BwdLds<D> &L) {
  filter_bwd_begin(d, P, L);
  for (int t = P.T - 1; t >= 0; --t) filter_bwd_step(d, P, S, G, ws, b, t, L);
  filter_bwd_end(d, G, b, L);
}

}  // namespace kvae
