// lgssm_n4.h — fused-phase sweeps for the headline shape n = 4 (dim z), p = 2 (dim a), any m <= 16:
// the same one-wavefront-per-sequence / LDS-tile model as lgssm_fwd.h / lgssm_bwd.h, but every phase
// does as much dependent arithmetic per lane as the 4x4 / 2x2 sizes allow, so a filter step is
// 4 phases (instead of 11), an RTS step 7 (9), and the backward reuses the gains saved by the forward
// (aux = Kt | S | J per step) instead of re-solving them: ~15 phases per step pair instead of ~38.
// At configs[1] the kernels are bound by the chain of dependent LDS round trips (one wavefront per CU),
// so phases, not flops or bytes, are the currency.
//
// Conventions that differ from the generic bodies:
//   * covariances are carried UNSYMMETRISED in LDS (F0 / M2); consumers read 0.5 (X[i,j] + X[j,i]) on the
//     fly, which is bit-identical to storing the symmetrised matrix (kalman_filter.py:101, :235);
//   * the 2x2 innovation system is solved per lane with the same partial-pivot LU recurrence lu_solve uses.
#pragma once
#include "lgssm_bwd.h"

namespace kvae {

// ---- register-blocked helpers: rows of 4 are moved LDS -> VGPR with one 16-byte read -----------------------------
struct alignas(16) kv4 { float x, y, z, w; };
struct M4 { kv4 r0, r1, r2, r3; };
KV_DEV kv4 ld4(const float *p) { return *reinterpret_cast<const kv4 *>(p); }
KV_DEV M4 ldm4(const float *p) { M4 m; m.r0 = ld4(p); m.r1 = ld4(p + 4); m.r2 = ld4(p + 8); m.r3 = ld4(p + 12); return m; }
KV_DEV float dot4(kv4 a, kv4 b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }
// fma chain in index order starting from acc (keeps the k = 0..3 summation order of the generic bodies)
KV_DEV float fma4(kv4 a, kv4 b, float acc) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc)))); }
KV_DEV kv4 col0(const M4 &m) { return kv4{m.r0.x, m.r1.x, m.r2.x, m.r3.x}; }
KV_DEV kv4 col1(const M4 &m) { return kv4{m.r0.y, m.r1.y, m.r2.y, m.r3.y}; }
KV_DEV kv4 col2(const M4 &m) { return kv4{m.r0.z, m.r1.z, m.r2.z, m.r3.z}; }
KV_DEV kv4 col3(const M4 &m) { return kv4{m.r0.w, m.r1.w, m.r2.w, m.r3.w}; }
KV_DEV M4 sym4(const M4 &m) {  // 0.5 (M + M^T), same arithmetic as KV_SYM4 element by element
  M4 s;
  s.r0 = kv4{0.5f * (m.r0.x + m.r0.x), 0.5f * (m.r0.y + m.r1.x), 0.5f * (m.r0.z + m.r2.x), 0.5f * (m.r0.w + m.r3.x)};
  s.r1 = kv4{0.5f * (m.r1.x + m.r0.y), 0.5f * (m.r1.y + m.r1.y), 0.5f * (m.r1.z + m.r2.y), 0.5f * (m.r1.w + m.r3.y)};
  s.r2 = kv4{0.5f * (m.r2.x + m.r0.z), 0.5f * (m.r2.y + m.r1.z), 0.5f * (m.r2.z + m.r2.z), 0.5f * (m.r2.w + m.r3.z)};
  s.r3 = kv4{0.5f * (m.r3.x + m.r0.w), 0.5f * (m.r3.y + m.r1.w), 0.5f * (m.r3.z + m.r2.w), 0.5f * (m.r3.w + m.r3.w)};
  return s;
}
// row-vector times matrix: (v M)[k] = sum_l v[l] M[l,k], l = 0..3 in order
KV_DEV kv4 vecmat(kv4 v, const M4 &m) {
  return kv4{fma4(v, col0(m), 0.f), fma4(v, col1(m), 0.f), fma4(v, col2(m), 0.f), fma4(v, col3(m), 0.f)};
}

template <int M>
struct alignas(16) N4Lds {
  // operands of the current step (every array starts on a 16-byte boundary)
  float A[16], Q[16], Bm[4 * ((M + 3) / 4 * 4)], C[8], R[4], u[(M + 3) / 4 * 4], y[4], mk[4];
  // filter state
  float mu[4], F0[16], mup[4], Sigp[16];
  float PCT[8], S[4], r[4], K[8], IKC[16], KR[8];
  // smoother
  float aug2[32], Xs[16], mus[4], M2[16], Sf[16], muf[4], dmu[4], Dm[16];
};

struct Sol2 { float x0, x1; };
// x = S^{-1} b for the 2x2 system, partial pivoting, same operation order as lu_solve()
KV_DEV Sol2 solve2(float s00, float s01, float s10, float s11, float b0, float b1) {
  const bool sw = fabsf(s10) > fabsf(s00);
  const float a00 = sw ? s10 : s00, a01 = sw ? s11 : s01, a10 = sw ? s00 : s10, a11 = sw ? s01 : s11;
  const float c0 = sw ? b1 : b0, c1 = sw ? b0 : b1;
  const float l = a10 * (1.0f / a00);
  const float u11 = fmaf(-l, a01, a11);
  const float y1 = fmaf(-l, c0, c1);
  Sol2 o;
  o.x1 = y1 / u11;
  o.x0 = fmaf(-a01, o.x1, c0) / a00;
  return o;
}

// x = Mt^{-1} b for a 4x4 system held in registers (rows m.r0..r3), partial pivoting by conditional row
// swaps (first maximum wins, as in lu_solve / getrf); every lane factorises redundantly, no LDS traffic.
KV_DEV kv4 solve4(M4 m, kv4 b) {
#define KV_SEL_SWAP(c_, a_, b_) { const float t_ = a_; a_ = c_ ? b_ : a_; b_ = c_ ? t_ : b_; }
#define KV_CSWAP(cond, ra, rb, ba, bb)                                                       \
  {                                                                                          \
    const bool c_ = (cond);                                                                  \
    KV_SEL_SWAP(c_, ra.x, rb.x) KV_SEL_SWAP(c_, ra.y, rb.y) KV_SEL_SWAP(c_, ra.z, rb.z)      \
    KV_SEL_SWAP(c_, ra.w, rb.w) KV_SEL_SWAP(c_, ba, bb)                                      \
  }
  // column 0
  {
    float best = fabsf(m.r0.x); int piv = 0;
    if (fabsf(m.r1.x) > best) { best = fabsf(m.r1.x); piv = 1; }
    if (fabsf(m.r2.x) > best) { best = fabsf(m.r2.x); piv = 2; }
    if (fabsf(m.r3.x) > best) { piv = 3; }
    KV_CSWAP(piv == 1, m.r0, m.r1, b.x, b.y) KV_CSWAP(piv == 2, m.r0, m.r2, b.x, b.z) KV_CSWAP(piv == 3, m.r0, m.r3, b.x, b.w)
    const float ri = 1.0f / m.r0.x;
    float l = m.r1.x * ri; m.r1.y = fmaf(-l, m.r0.y, m.r1.y); m.r1.z = fmaf(-l, m.r0.z, m.r1.z); m.r1.w = fmaf(-l, m.r0.w, m.r1.w); b.y = fmaf(-l, b.x, b.y);
    l = m.r2.x * ri; m.r2.y = fmaf(-l, m.r0.y, m.r2.y); m.r2.z = fmaf(-l, m.r0.z, m.r2.z); m.r2.w = fmaf(-l, m.r0.w, m.r2.w); b.z = fmaf(-l, b.x, b.z);
    l = m.r3.x * ri; m.r3.y = fmaf(-l, m.r0.y, m.r3.y); m.r3.z = fmaf(-l, m.r0.z, m.r3.z); m.r3.w = fmaf(-l, m.r0.w, m.r3.w); b.w = fmaf(-l, b.x, b.w);
  }
  // column 1
  {
    float best = fabsf(m.r1.y); int piv = 1;
    if (fabsf(m.r2.y) > best) { best = fabsf(m.r2.y); piv = 2; }
    if (fabsf(m.r3.y) > best) { piv = 3; }
    KV_CSWAP(piv == 2, m.r1, m.r2, b.y, b.z) KV_CSWAP(piv == 3, m.r1, m.r3, b.y, b.w)
    const float ri = 1.0f / m.r1.y;
    float l = m.r2.y * ri; m.r2.z = fmaf(-l, m.r1.z, m.r2.z); m.r2.w = fmaf(-l, m.r1.w, m.r2.w); b.z = fmaf(-l, b.y, b.z);
    l = m.r3.y * ri; m.r3.z = fmaf(-l, m.r1.z, m.r3.z); m.r3.w = fmaf(-l, m.r1.w, m.r3.w); b.w = fmaf(-l, b.y, b.w);
  }
  // column 2
  {
    KV_CSWAP(fabsf(m.r3.z) > fabsf(m.r2.z), m.r2, m.r3, b.z, b.w)
    const float l = m.r3.z * (1.0f / m.r2.z);
    m.r3.w = fmaf(-l, m.r2.w, m.r3.w); b.w = fmaf(-l, b.z, b.w);
  }
#undef KV_CSWAP
#undef KV_SEL_SWAP
  kv4 x;
  x.w = b.w / m.r3.w;
  x.z = fmaf(-m.r2.w, x.w, b.z) / m.r2.z;
  x.y = fmaf(-m.r1.w, x.w, fmaf(-m.r1.z, x.z, b.y)) / m.r1.y;
  x.x = fmaf(-m.r0.w, x.w, fmaf(-m.r0.z, x.z, fmaf(-m.r0.y, x.y, b.x))) / m.r0.x;
  return x;
}

#define KV_SYM4(X, i, j) (0.5f * ((X)[(i) * 4 + (j)] + (X)[(j) * 4 + (i)]))

// aux record per (b,t): K unmasked [4,2] | S [2,2] | J [4,4]
#define KV_AUX_N4 28

template <class D>
KV_DEV void filter_sweep_n4(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b,
                            N4Lds<D::MMAX> &L) {
  constexpr int n = 4, p = 2, m = D::MMAX, nn = 16;
  static_assert(m == 4, "register-blocked path is written for m = 4");
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  copy_in(L.mu, P.mu0 + (int64_t)b * P.mu0_sb, n);
  copy_in(L.F0, P.Sigma0 + (int64_t)b * P.Sigma0_sb, nn);
  copy_in(L.R, P.R, p * p);
  StepOperands<D> pf;
  operands_issue(d, P, b, 0, pf);
  for (int t = 0; t < T; ++t) {
    operands_commit(d, P, pf, L);
    if (t + 1 < T) operands_issue(d, P, b, t + 1, pf);
    KV_SYNC();
    const bool first = (t == 0);  // Sigma0 is used as given; later covariances are read symmetrised
    // ---- F1: Sig_p = (A Sig) A^T + Q ; mu_p = A mu + B u ; stream out Sig_f of the previous step ----------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      const M4 F = ldm4(L.F0);
      const M4 Sg = first ? F : sym4(F);
      const kv4 as = vecmat(ld4(L.A + 4 * i), Sg);          // (A Sig)[i,:]
      const float v = fma4(as, ld4(L.A + 4 * j), 0.f) + L.Q[e];
      L.Sigp[e] = v;
      S.Sigmas_pred[(bT + t) * nn + e] = v;
      if (!first) S.Sigmas_filt[(bT + t - 1) * nn + e] = KV_SYM4(L.F0, i, j);
    }
    KV_PAR(i, n) {
      const float acc = fma4(ld4(L.Bm + 4 * i), ld4(L.u), fma4(ld4(L.A + 4 * i), ld4(L.mu), 0.f));
      L.mup[i] = acc;
      S.mus_pred[(bT + t) * n + i] = acc;
    }
    KV_SYNC();
    // ---- F2: PCT = Sig_p C^T ; S = sym(C Sig_p C^T + R) ; r = y - C mu_p -----------------------------------------
    KV_PAR(e, n * p) {
      const int j = e >> 1, i = e & 1;  // PCT[j,i]
      L.PCT[e] = fma4(ld4(L.Sigp + 4 * j), ld4(L.C + 4 * i), 0.f);
    }
    KV_PAR(e, p * p) {
      const int a = e >> 1, c = e & 1;
      const M4 Sp = ldm4(L.Sigp);
      const kv4 ca = ld4(L.C + 4 * a), cc = ld4(L.C + 4 * c);
      const float s1 = fma4(vecmat(ca, Sp), cc, 0.f), s2 = fma4(vecmat(cc, Sp), ca, 0.f);
      L.S[e] = 0.5f * ((s1 + L.R[a * 2 + c]) + (s2 + L.R[c * 2 + a]));
    }
    KV_PAR(i, p) { L.r[i] = L.y[i] - fma4(ld4(L.C + 4 * i), ld4(L.mup), 0.f); }
    KV_SYNC();
    // ---- F3: K = PCT S^{-1} (per-lane 2x2 solve) ; IKC = I - K C ; KR = K R ; mu_f = mu_p + K r --------------------
    const float mk = L.mk[0];
    float *aux = S.aux ? S.aux + (bT + t) * KV_AUX_N4 : nullptr;
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      const kv4 s = ld4(L.S);
      const Sol2 k = solve2(s.x, s.y, s.z, s.w, L.PCT[i * 2], L.PCT[i * 2 + 1]);
      float acc = 0.f;
      acc = fmaf(mk * k.x0, L.C[j], acc);
      acc = fmaf(mk * k.x1, L.C[4 + j], acc);
      L.IKC[e] = (i == j ? 1.0f : 0.0f) - acc;
      if (j < 2) {  // lanes (i,0),(i,1) also publish K[i,j], KR[i,j]
        const float kij = j == 0 ? k.x0 : k.x1;
        L.K[i * 2 + j] = mk * kij;
        float kr = 0.f;
        kr = fmaf(mk * k.x0, L.R[j], kr);
        kr = fmaf(mk * k.x1, L.R[2 + j], kr);
        L.KR[i * 2 + j] = kr;
        if (aux) aux[i * 2 + j] = kij;
      } else if (j == 2) {
        float acc2 = L.mup[i];
        acc2 = fmaf(mk * k.x0, L.r[0], acc2);
        acc2 = fmaf(mk * k.x1, L.r[1], acc2);
        L.mu[i] = acc2;   // mu is not read in this phase
        S.mus_filt[(bT + t) * n + i] = acc2;
      } else if (aux) {   // j == 3: save S (4 values) once per step from lanes (0..3, 3)
        aux[8 + i] = L.S[i];
      }
    }
    KV_SYNC();
    // ---- F4: F0 = (IKC Sig_p) IKC^T + KR K^T   (symmetrisation is done by the readers) -------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      const kv4 t1 = vecmat(ld4(L.IKC + 4 * i), ldm4(L.Sigp));   // (IKC Sig_p)[i,:]
      const float acc = fma4(t1, ld4(L.IKC + 4 * j), 0.f);
      float acc2 = 0.f;
      acc2 = fmaf(L.KR[i * 2], L.K[j * 2], acc2);
      acc2 = fmaf(L.KR[i * 2 + 1], L.K[j * 2 + 1], acc2);
      L.F0[e] = acc + acc2;
    }
    KV_SYNC();
  }
  KV_PAR(e, nn) {
    const int i = e >> 2, j = e & 3;
    S.Sigmas_filt[(bT + T - 1) * nn + e] = KV_SYM4(L.F0, i, j);
  }
  KV_SYNC();
}

// RTS sweep, n = 4: reads the filtered/predicted stacks back from global memory (written by this wave).
// The 4x4 system of J = Sig_f A^T Sig_p^{-1} is solved per lane in registers (one right-hand side each).
template <class D>
KV_DEV void rts_sweep_n4(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, N4Lds<D::MMAX> &L) {
  constexpr int n = 4, nn = 16;
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  copy_in(L.mus, S.mus_filt + (bT + T - 1) * n, n);
  copy_in(L.M2, S.Sigmas_filt + (bT + T - 1) * nn, nn);   // symmetric already
  KV_SYNC();
  copy_out(S.mus_smooth + (bT + T - 1) * n, L.mus, n);
  copy_out(S.Sigmas_smooth + (bT + T - 1) * nn, L.M2, nn);
  Prefetch<16> pfS, pfP, pfA;
  Prefetch<4> pfm, pfq;
  if (T >= 2) {
    pfS.issue(S.Sigmas_filt + (bT + T - 2) * nn, nn);
    pfP.issue(S.Sigmas_pred + (bT + T - 1) * nn, nn);
    pfA.issue(stack_at(P.A, b, T - 1), nn);
    pfm.issue(S.mus_filt + (bT + T - 2) * n, n);
    pfq.issue(S.mus_pred + (bT + T - 1) * n, n);
  }
  for (int t = T - 2; t >= 0; --t) {
    pfS.commit(L.Sf, nn);
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
    KV_SYNC();
    float *aux = S.aux ? S.aux + (bT + t) * KV_AUX_N4 : nullptr;
    // ---- R1: row i of J solves Sig_p^T x = (Sig_f A^T)[i,:]^T (lanes (i,0)); D = Sig_s(t+1) - Sig_p; stream out Sig_s(t+1)
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      const float ss = KV_SYM4(L.M2, i, j);
      L.Dm[e] = ss - L.Sigp[e];
      if (t < T - 2) S.Sigmas_smooth[(bT + t + 1) * nn + e] = ss;
      if (j == 0) {
        const M4 Sp = ldm4(L.Sigp);
        M4 SpT; SpT.r0 = col0(Sp); SpT.r1 = col1(Sp); SpT.r2 = col2(Sp); SpT.r3 = col3(Sp);
        const kv4 sf = ld4(L.Sf + 4 * i);
        const M4 Am = ldm4(L.A);
        // W[i,k] = sum_l Sig_f[i,l] A[k,l]
        const kv4 wrow = kv4{fma4(sf, Am.r0, 0.f), fma4(sf, Am.r1, 0.f), fma4(sf, Am.r2, 0.f), fma4(sf, Am.r3, 0.f)};
        const kv4 jr = solve4(SpT, wrow);   // J[i,:]
        L.Xs[0 * 4 + i] = jr.x; L.Xs[1 * 4 + i] = jr.y; L.Xs[2 * 4 + i] = jr.z; L.Xs[3 * 4 + i] = jr.w;   // Xs = J^T
        if (aux) { aux[12 + 4 * i] = jr.x; aux[12 + 4 * i + 1] = jr.y; aux[12 + 4 * i + 2] = jr.z; aux[12 + 4 * i + 3] = jr.w; }
      } else if (j == 1) {
        L.dmu[i] = L.mus[i] - L.mup[i];
      }
    }
    KV_SYNC();
    // ---- R2: mu_s = mu_f + J dmu ; M2 = Sig_f + (J D) J^T (unsymmetrised) ------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      const M4 Xt = ldm4(L.Xs);                      // Xt rows = columns of J: Xt[l][i] = J[i,l]
      const kv4 ji = kv4{L.Xs[i], L.Xs[4 + i], L.Xs[8 + i], L.Xs[12 + i]};   // J[i,:]
      const kv4 jj = kv4{L.Xs[j], L.Xs[4 + j], L.Xs[8 + j], L.Xs[12 + j]};   // J[j,:]
      (void)Xt;
      const kv4 tj = vecmat(ji, ldm4(L.Dm));         // (J D)[i,:]
      L.M2[e] = L.Sf[e] + fma4(tj, jj, 0.f);         // M2 was last read in R1 (a phase ago)
      if (j == 0) {
        const float acc = fma4(ji, ld4(L.dmu), L.muf[i]);
        L.mu[i] = acc;  // staging
        S.mus_smooth[(bT + t) * n + i] = acc;
      }
    }
    KV_SYNC();
    KV_PAR(i, n) { L.mus[i] = L.mu[i]; }
  }
  KV_SYNC();
  if (T >= 2) {
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      S.Sigmas_smooth[bT * nn + e] = KV_SYM4(L.M2, i, j);
    }
  }
}

}  // namespace kvae

// ================================================================================================
// backward, n = 4, p = 2: adjoint sweeps reusing the gains saved by the forward (aux = K | S | J)
// ================================================================================================
namespace kvae {

template <int M>
struct alignas(16) N4BwdLds {
  float A[16], Q[16], Bm[4 * ((M + 3) / 4 * 4)], C[8], R[4], u[(M + 3) / 4 * 4], y[4], mk[4];
  float mu[4], Sig[16], mup[4], Sigp[16];
  float K[8], S[4], J[16];                     // saved gains of the step (K unmasked)
  // filter adjoints
  float gmu[4], gSig[16];                      // carried adjoint of the filtered belief (from step t+1)
  float gmuT[4], G[16], IKC[16], r[4], gr[4];
  float gIKC[16], gSp[16], GK[8], gK[8], gC1[8], Z[8], gS0[4], gmp[4];
  // smoother adjoints
  float Sf[16], mus[4], Sigs[16];
  float gsm[4], gsS[16];                       // carried adjoint of the smoothed belief at t
  float gM[16], gD[16], gdm[4], aug2[32], gR[16];
};

template <class D>
KV_DEV void rts_bwd_sweep_n4(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                             const kvae_lgssm_states &U, const kvae_lgssm_input_grads &G, float *ws, int b,
                             N4BwdLds<D::MMAX> &L) {
  constexpr int n = 4, nn = 16, rec = 2 * (n + nn);
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  float *w = ws + bT * rec;
  KV_PAR(i, n) {
    L.gsm[i] = opt_load(U.mus_smooth, bT * n + i);
    w[n + nn + i] = opt_load(U.mus_pred, bT * n + i);
  }
  KV_PAR(e, nn) {
    L.gsS[e] = opt_load(U.Sigmas_smooth, bT * nn + e);
    w[n + nn + n + e] = opt_load(U.Sigmas_pred, bT * nn + e);
    gstack_at(G.gA, b, 0)[e] = 0.0f;
  }
  KV_SYNC();
  for (int t = 0; t + 1 < T; ++t) {
    const int64_t q = bT + t;
    copy_in(L.Sf, S.Sigmas_filt + q * nn, nn);
    copy_in(L.Sigp, S.Sigmas_pred + (q + 1) * nn, nn);
    copy_in(L.A, stack_at(P.A, b, t + 1), nn);
    copy_in(L.mup, S.mus_pred + (q + 1) * n, n);
    copy_in(L.mus, S.mus_smooth + (q + 1) * n, n);
    copy_in(L.Sigs, S.Sigmas_smooth + (q + 1) * nn, nn);
    copy_in(L.J, S.aux + q * KV_AUX_N4 + 12, nn);
    KV_SYNC();
    // ---- B1: gM, Y1 = gM J (row i and column j per lane), gJ -> aug2 = [Sig_p | gJ^T], gD, gdm -------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      float gj = 0.f, gd = 0.f;
      for (int k = 0; k < n; ++k) {
        float y_ik = 0.f, y_kj = 0.f;  // Y1[i,k], Y1[k,j]
        for (int l = 0; l < n; ++l) {
          y_ik = fmaf(KV_SYM4(L.gsS, i, l), L.J[l * 4 + k], y_ik);
          y_kj = fmaf(KV_SYM4(L.gsS, k, l), L.J[l * 4 + j], y_kj);
        }
        const float dsym = (L.Sigs[j * 4 + k] - L.Sigp[j * 4 + k]) + (L.Sigs[k * 4 + j] - L.Sigp[k * 4 + j]);
        gj = fmaf(y_ik, dsym, gj);
        gd = fmaf(L.J[k * 4 + i], y_kj, gd);
      }
      gj = fmaf(L.gsm[i], L.mus[j] - L.mup[j], gj);
      L.aug2[e] = gj;          // gJ, row-major (first 16 floats of aug2)
      L.gD[e] = gd;
      L.gM[e] = KV_SYM4(L.gsS, i, j);
    }
    KV_PAR(i, n) {
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.J[k * 4 + i], L.gsm[k], acc);
      L.gdm[i] = acc;
    }
    KV_SYNC();
    // ---- B2: gR = Sig_p^{-1} gJ^T: column c of gR solves Sig_p x = gJ[c,:]^T, one system per lane, in registers ----------
    KV_PAR(c, n) {
      const kv4 x = solve4(ldm4(L.Sigp), ld4(L.aug2 + 4 * c));
      L.gR[0 * 4 + c] = x.x; L.gR[1 * 4 + c] = x.y; L.gR[2 * 4 + c] = x.z; L.gR[3 * 4 + c] = x.w;
    }
    KV_SYNC();
    // ---- B7: hand-offs to the filter sweep, smoother share of gA, carried adjoint of the smoothed belief at t+1 ---------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      float acc = 0.f, gp = 0.f, ga = 0.f;
      for (int k = 0; k < n; ++k) {
        acc = fmaf(L.gR[k * 4 + i], L.A[k * 4 + j], acc);     // (gW A)[i,j]
        gp = fmaf(L.J[k * 4 + i], L.gR[j * 4 + k], gp);       // (J^T gW)[i,j]
        ga = fmaf(L.gR[i * 4 + k], L.Sf[k * 4 + j], ga);      // (gW^T Sig_f)[i,j]
      }
      w[n + e] = opt_load(U.Sigmas_filt, q * nn + e) + L.gM[e] + acc;
      w[rec + n + nn + n + e] = opt_load(U.Sigmas_pred, (q + 1) * nn + e) - L.gD[e] - gp;
      gstack_at(G.gA, b, t + 1)[e] = ga;
      L.gsS[e] = opt_load(U.Sigmas_smooth, (q + 1) * nn + e) + L.gD[e];   // gsS is not read in this phase
    }
    KV_PAR(i, n) {
      w[i] = opt_load(U.mus_filt, q * n + i) + L.gsm[i];
      w[rec + n + nn + i] = opt_load(U.mus_pred, (q + 1) * n + i) - L.gdm[i];
      L.gsm[i] = opt_load(U.mus_smooth, (q + 1) * n + i) + L.gdm[i];      // same lane reads then writes gsm[i]
    }
    KV_SYNC();
    w += rec;
  }
  const int64_t q = bT + T - 1;
  KV_PAR(e, nn) { w[n + e] = opt_load(U.Sigmas_filt, q * nn + e) + L.gsS[e]; }
  KV_PAR(i, n) { w[i] = opt_load(U.mus_filt, q * n + i) + L.gsm[i]; }
  KV_SYNC();
}

template <class D>
KV_DEV void filter_bwd_sweep_n4(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                                const kvae_lgssm_input_grads &G, const float *ws, int b, N4BwdLds<D::MMAX> &L) {
  constexpr int n = 4, p = 2, m = D::MMAX, nn = 16, rec = 2 * (n + nn);
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  copy_in(L.R, P.R, p * p);
  KV_LANE0 { L.mk[0] = 1.0f; }
  KV_PAR(i, n) { L.gmu[i] = 0.0f; }
  KV_PAR(e, nn) { L.gSig[e] = 0.0f; }
  KV_SYNC();
  for (int t = T - 1; t >= 0; --t) {
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
    copy_in(L.K, S.aux + q * KV_AUX_N4, 8);
    copy_in(L.S, S.aux + q * KV_AUX_N4 + 8, 4);
    KV_SYNC();
    const float mk = L.mk[0];
    // ---- P1: totals of the incoming adjoints, G = sym(gSig), IKC, r, gr ---------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      L.G[e] = 0.5f * ((L.gSig[e] + w[n + e]) + (L.gSig[j * 4 + i] + w[n + j * 4 + i]));
      float acc = 0.f;
      for (int c = 0; c < p; ++c) acc = fmaf(mk * L.K[i * 2 + c], L.C[c * 4 + j], acc);
      L.IKC[e] = (i == j ? 1.0f : 0.0f) - acc;
    }
    KV_PAR(i, n) { L.gmuT[i] = L.gmu[i] + w[i]; }
    KV_PAR(c, p) {
      float acc = 0.f, g = 0.f;
      for (int k = 0; k < n; ++k) {
        acc = fmaf(L.C[c * 4 + k], L.mup[k], acc);
        g = fmaf(mk * L.K[k * 2 + c], L.gmu[k] + w[k], g);
      }
      L.r[c] = L.y[c] - acc;
      L.gr[c] = g;
    }
    KV_SYNC();
    // ---- P2: X1 = G IKC (row i, column j per lane) -> gIKC, gSp0 ; GK = G K ------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      float gi = 0.f, gs = 0.f;
      for (int k = 0; k < n; ++k) {
        float x_ik = 0.f, x_kj = 0.f;
        for (int l = 0; l < n; ++l) {
          x_ik = fmaf(L.G[i * 4 + l], L.IKC[l * 4 + k], x_ik);
          x_kj = fmaf(L.G[k * 4 + l], L.IKC[l * 4 + j], x_kj);
        }
        gi = fmaf(x_ik, L.Sigp[j * 4 + k] + L.Sigp[k * 4 + j], gi);
        gs = fmaf(L.IKC[k * 4 + i], x_kj, gs);
      }
      L.gIKC[e] = gi;
      L.gSp[e] = gs + w[n + nn + n + e];
      if (j < p) {
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(L.G[i * 4 + k], mk * L.K[k * 2 + j], acc);
        L.GK[i * 2 + j] = acc;
      }
    }
    KV_SYNC();
    // ---- P3: gK, gC1 ------------------------------------------------------------------------------------------------------------
    KV_PAR(e, n * p) {
      const int i = e >> 1, c = e & 1;
      float acc = 0.f;
      for (int k = 0; k < p; ++k) acc = fmaf(L.GK[i * 2 + k], L.R[c * 2 + k] + L.R[k * 2 + c], acc);
      for (int k = 0; k < n; ++k) acc = fmaf(-L.gIKC[i * 4 + k], L.C[c * 4 + k], acc);
      acc = fmaf(L.gmuT[i], L.r[c], acc);
      L.gK[e] = acc;
      const int c2 = e >> 2, j2 = e & 3;   // the same 8 lanes also produce gC1[c2,j2] = -(K^T gIKC)
      float a2 = 0.f;
      for (int k = 0; k < n; ++k) a2 = fmaf(mk * L.K[k * 2 + c2], L.gIKC[k * 4 + j2], a2);
      L.gC1[e] = -a2;
    }
    KV_SYNC();
    // ---- P4: Z = solve(S^T, mask gK^T), one column per lane ------------------------------------------------------------------------
    KV_PAR(j, n) {
      const Sol2 z = solve2(L.S[0], L.S[2], L.S[1], L.S[3], mk * L.gK[j * 2], mk * L.gK[j * 2 + 1]);
      L.Z[j] = z.x0;
      L.Z[4 + j] = z.x1;
    }
    KV_SYNC();
    // ---- P5: gS0 = sym(-Z Kt^T), Kt = unmasked K^T ---------------------------------------------------------------------------------------
    KV_PAR(e, p * p) {
      const int a = e >> 1, c = e & 1;
      float s1 = 0.f, s2 = 0.f;
      for (int k = 0; k < n; ++k) {
        s1 = fmaf(L.Z[a * 4 + k], L.K[k * 2 + c], s1);
        s2 = fmaf(L.Z[c * 4 + k], L.K[k * 2 + a], s2);
      }
      L.gS0[e] = -0.5f * (s1 + s2);
    }
    KV_SYNC();
    // ---- P6: gSp (final) ; gC ; gmp ; gY -------------------------------------------------------------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      float acc = L.gSp[e];
      for (int k = 0; k < p; ++k) {
        float gcp = 0.f;  // gCP[k,j] = (gS0 C)[k,j]
        for (int c = 0; c < p; ++c) gcp = fmaf(L.gS0[k * 2 + c], L.C[c * 4 + j], gcp);
        acc = fmaf(L.Z[k * 4 + i], L.C[k * 4 + j], acc);
        acc = fmaf(L.C[k * 4 + i], gcp, acc);
      }
      L.gIKC[e] = acc;   // final gSp parked in gIKC (dead after P3) so that gSp is not read and written in one phase
      if (G.gQ.ptr) gstack_at(G.gQ, b, t)[e] = acc;
    }
    KV_PAR(e, p * n) {
      const int c = e >> 2, j = e & 3;
      float acc = L.gC1[e];
      for (int k = 0; k < n; ++k) {
        float gcp = 0.f;  // gCP[c,k]
        for (int cc = 0; cc < p; ++cc) gcp = fmaf(L.gS0[c * 2 + cc], L.C[cc * 4 + k], gcp);
        acc = fmaf(L.Z[c * 4 + k], L.Sigp[k * 4 + j], acc);
        acc = fmaf(gcp, L.Sigp[j * 4 + k], acc);
      }
      for (int k = 0; k < p; ++k) {
        float cp = 0.f;  // CP[k,j] = (C Sig_p)[k,j]
        for (int l = 0; l < n; ++l) cp = fmaf(L.C[k * 4 + l], L.Sigp[l * 4 + j], cp);
        acc = fmaf(L.gS0[c * 2 + k], cp, acc);
      }
      acc = fmaf(-L.gr[c], L.mup[j], acc);
      gstack_at(G.gC, b, t)[e] = acc;
    }
    KV_PAR(i, n) {
      float acc = L.gmuT[i] + w[n + nn + i];
      for (int k = 0; k < p; ++k) acc = fmaf(-L.C[k * 4 + i], L.gr[k], acc);
      L.gmp[i] = acc;
    }
    KV_PAR(c, p) { G.gY[q * p + c] = L.gr[c]; }
    KV_SYNC();
    // ---- P7: gA, carried adjoints of step t-1, gB, gU --------------------------------------------------------------------------------------------
    const float *gSpF = L.gIKC;
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      float acc = gstack_at(G.gA, b, t)[e];
      float gs = 0.f;
      for (int k = 0; k < n; ++k) {
        float as_kj = 0.f, gas_ik = 0.f, gas_kj = 0.f;  // (A Sig)[k,j], (gSp A)[i,k], (gSp A)[k,j]
        for (int l = 0; l < n; ++l) {
          as_kj = fmaf(L.A[k * 4 + l], L.Sig[l * 4 + j], as_kj);
          gas_ik = fmaf(gSpF[i * 4 + l], L.A[l * 4 + k], gas_ik);
          gas_kj = fmaf(gSpF[k * 4 + l], L.A[l * 4 + j], gas_kj);
        }
        acc = fmaf(gSpF[k * 4 + i], as_kj, acc);
        acc = fmaf(gas_ik, L.Sig[j * 4 + k], acc);
        gs = fmaf(L.A[k * 4 + i], gas_kj, gs);
      }
      acc = fmaf(L.gmp[i], L.mu[j], acc);
      gstack_at(G.gA, b, t)[e] = acc;
      L.gSig[e] = gs;   // gSig was last read in P1
    }
    KV_PAR(i, n) {
      float acc = 0.f;
      for (int k = 0; k < n; ++k) acc = fmaf(L.A[k * 4 + i], L.gmp[k], acc);
      L.gmu[i] = acc;   // gmu was last read in P1
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
  if (G.g_mu0) copy_out(G.g_mu0 + (int64_t)b * n, L.gmu, n);
  if (G.g_Sigma0) copy_out(G.g_Sigma0 + (int64_t)b * nn, L.gSig, nn);
}

}  // namespace kvae
