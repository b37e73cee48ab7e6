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
  // operands (every array starts on a 16-byte boundary; *T = transposed copy so that columns are 16-byte rows too)
  float A[16], AT[16], Q[16], Bm[4 * ((M + 3) / 4 * 4)], C[8], R[4], u[(M + 3) / 4 * 4], y[4], mk[4];
  float mu[4], Sig[16], SigT[16], mup[4], Sigp[16], SpT[16], Sp2[16];
  float K[8], KT[8], S[4], J[16], JT[16];      // saved gains of the step (K unmasked)
  // filter adjoints
  float gmu[4], gSig[16];                      // carried adjoint of the filtered belief (from step t+1)
  float gmuT[4], G[16], IKC[16], IKCT[16], r[4], gr[4];
  float gIKC[16], gIKCT[16], gSp[16], gSpF[16], gSpFT[16], GK[8], gK[8], gC1[8], Z[8], gS0[4], gmp[4];
  // smoother adjoints
  float Sf[16], mus[4], D2[16];
  float gsm[4], gM[16];                        // carried adjoint of the smoothed belief at t (gM symmetrised)
  float gD[16], gdm[4], gJ[16], gR[16], gRT[16];
};

KV_DEV void st4(float *p, kv4 v) { *reinterpret_cast<kv4 *>(p) = v; }

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
    const int eT = (e & 3) * 4 + (e >> 2);
    L.gM[e] = 0.5f * (opt_load(U.Sigmas_smooth, bT * nn + e) + opt_load(U.Sigmas_smooth, bT * nn + eT));
    w[n + nn + n + e] = opt_load(U.Sigmas_pred, bT * nn + e);
    gstack_at(G.gA, b, 0)[e] = 0.0f;
  }
  KV_SYNC();
  // per-lane register prefetch of step t+1's operands and upstream gradients while step t computes
  #define KV_S16 KV_PF_SLOTS(16)
  struct RtsPf { float sf[KV_S16], sp[KV_S16], d2[KV_S16], a[KV_S16], jv[KV_S16], mp[KV_S16], ms[KV_S16], u_sf[KV_S16], u_sp[KV_S16],
                 u_ss[KV_S16], u_ssT[KV_S16], u_mf[KV_S16], u_mp[KV_S16], u_ms[KV_S16]; } pf;
  auto issue = [&](int t) {
    const int64_t q = bT + t;
    for (int s_ = 0; s_ < KV_S16; ++s_) {
    const int e = (s_ * KV_LANES + KV_LANE) & 15, eT = (e & 3) * 4 + (e >> 2), i = e & 3;
    const float *Sp = S.Sigmas_pred + (q + 1) * nn, *Ss = S.Sigmas_smooth + (q + 1) * nn;
    pf.sp[s_] = Sp[e];
    pf.sf[s_] = S.Sigmas_filt[q * nn + e];
    pf.d2[s_] = (Ss[e] - pf.sp[s_]) + (Ss[eT] - Sp[eT]);
    pf.a[s_] = stack_at(P.A, b, t + 1)[e];
    pf.jv[s_] = S.aux[q * KV_AUX_N4 + 12 + e];
    pf.mp[s_] = S.mus_pred[(q + 1) * n + i];
    pf.ms[s_] = S.mus_smooth[(q + 1) * n + i];
    pf.u_sf[s_] = opt_load(U.Sigmas_filt, q * nn + e);
    pf.u_sp[s_] = opt_load(U.Sigmas_pred, (q + 1) * nn + e);
    pf.u_ss[s_] = opt_load(U.Sigmas_smooth, (q + 1) * nn + e);
    pf.u_ssT[s_] = opt_load(U.Sigmas_smooth, (q + 1) * nn + eT);
    pf.u_mf[s_] = opt_load(U.mus_filt, q * n + i);
    pf.u_mp[s_] = opt_load(U.mus_pred, (q + 1) * n + i);
    pf.u_ms[s_] = opt_load(U.mus_smooth, (q + 1) * n + i);
    }
  };
  if (T >= 2) issue(0);
  for (int t = 0; t + 1 < T; ++t) {
    const int64_t q = bT + t;
    // ---- commit the prefetched step (with transposed copies), then prefetch the next one ------------------------------------
    const RtsPf cur = pf;
    KV_PAR(e, nn) {
      const int eT = (e & 3) * 4 + (e >> 2);
      L.Sf[e] = cur.sf[e / KV_LANES];
      L.Sigp[e] = cur.sp[e / KV_LANES];
      L.D2[e] = cur.d2[e / KV_LANES];
      L.A[e] = cur.a[e / KV_LANES];
      L.AT[eT] = cur.a[e / KV_LANES];
      L.J[e] = cur.jv[e / KV_LANES];
      L.JT[eT] = cur.jv[e / KV_LANES];
      if (e < n) { L.mup[e] = cur.mp[e / KV_LANES]; L.mus[e] = cur.ms[e / KV_LANES]; }
    }
    if (t + 2 < T) issue(t + 1);
    KV_SYNC();
    // ---- B1: Y1 = gM J (row i, column j per lane) -> gJ, gD ; gdm = J^T gsm ------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      const M4 Jm = ldm4(L.J), gMm = ldm4(L.gM);
      const kv4 y_row = vecmat(ld4(L.gM + 4 * i), Jm);                 // Y1[i,:]
      const kv4 jcj = ld4(L.JT + 4 * j), jci = ld4(L.JT + 4 * i);      // J[:,j], J[:,i]
      const kv4 y_col = kv4{fma4(gMm.r0, jcj, 0.f), fma4(gMm.r1, jcj, 0.f), fma4(gMm.r2, jcj, 0.f), fma4(gMm.r3, jcj, 0.f)};
      L.gJ[e] = fmaf(L.gsm[i], L.mus[j] - L.mup[j], fma4(y_row, ld4(L.D2 + 4 * j), 0.f));
      L.gD[e] = fma4(jci, y_col, 0.f);
      if (j == 0) L.gdm[i] = fma4(jci, ld4(L.gsm), 0.f);
    }
    KV_SYNC();
    // ---- B2: gR = Sig_p^{-1} gJ^T, one 4x4 system per lane (registers) ; gRT keeps gR[:,c] as a row ----------------------------
    KV_PAR(c, n) {
      const kv4 x = solve4(ldm4(L.Sigp), ld4(L.gJ + 4 * c));
      L.gR[0 * 4 + c] = x.x; L.gR[1 * 4 + c] = x.y; L.gR[2 * 4 + c] = x.z; L.gR[3 * 4 + c] = x.w;
      st4(L.gRT + 4 * c, x);
    }
    KV_SYNC();
    // ---- B7: hand-offs to the filter sweep, smoother share of gA, carried adjoint of the smoothed belief at t+1 ---------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3, eT = j * 4 + i;
      const float acc = fma4(ld4(L.gRT + 4 * i), ld4(L.AT + 4 * j), 0.f);   // (gW A)[i,j]
      const float gp = fma4(ld4(L.JT + 4 * i), ld4(L.gR + 4 * j), 0.f);     // (J^T gW)[i,j]
      const float ga = fma4(ld4(L.gR + 4 * i), ld4(L.Sf + 4 * j), 0.f);     // (gW^T Sig_f)[i,j]  (Sig_f symmetric)
      const float gm = L.gM[e];
      w[n + e] = cur.u_sf[e / KV_LANES] + gm + acc;
      w[rec + n + nn + n + e] = cur.u_sp[e / KV_LANES] - L.gD[e] - gp;
      gstack_at(G.gA, b, t + 1)[e] = ga;
      L.gJ[e] = 0.5f * ((cur.u_ss[e / KV_LANES] + L.gD[e]) + (cur.u_ssT[e / KV_LANES] + L.gD[eT]));   // next gM, staged in gJ
    }
    KV_PAR(i, n) {
      w[i] = cur.u_mf[i / KV_LANES] + L.gsm[i];
      w[rec + n + nn + i] = cur.u_mp[i / KV_LANES] - L.gdm[i];
      L.gsm[i] = cur.u_ms[i / KV_LANES] + L.gdm[i];      // same lane reads then writes gsm[i]
    }
    KV_SYNC();
    KV_PAR(e, nn) { L.gM[e] = L.gJ[e]; }
    w += rec;
  }
  KV_SYNC();
  const int64_t q = bT + T - 1;
  KV_PAR(e, nn) { w[n + e] = opt_load(U.Sigmas_filt, q * nn + e) + L.gM[e]; }
  KV_PAR(i, n) { w[i] = opt_load(U.mus_filt, q * n + i) + L.gsm[i]; }
  KV_SYNC();
}

template <class D>
KV_DEV void filter_bwd_sweep_n4(const D d, const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                                const kvae_lgssm_input_grads &G, const float *ws, int b, N4BwdLds<D::MMAX> &L) {
  constexpr int n = 4, p = 2, m = D::MMAX, nn = 16, rec = 2 * (n + nn);
  static_assert(m == 4, "register-blocked path is written for m = 4");
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  copy_in(L.R, P.R, p * p);
  KV_PAR(i, n) { L.gmu[i] = 0.0f; }
  KV_PAR(e, nn) { L.gSig[e] = 0.0f; }
  KV_SYNC();
  // per-lane register prefetch of step t-1's operands, saved gains and handed-off adjoints while step t computes
  struct FbPf { float a[KV_S16], sg[KV_S16], sp[KV_S16], spT[KV_S16], bm[KV_S16], c[KV_S16], kv[KV_S16], s4[KV_S16], mu[KV_S16], mup[KV_S16],
                u[KV_S16], y[KV_S16], mkv[KV_S16], w_sf[KV_S16], w_sfT[KV_S16], w_sp[KV_S16], w_mf[KV_S16], w_mp[KV_S16]; } pf;
  auto issue = [&](int t) {
    const int64_t q = bT + t;
    const float *w = ws + q * rec;
    for (int s_ = 0; s_ < KV_S16; ++s_) {
    const int e = (s_ * KV_LANES + KV_LANE) & 15, eT = (e & 3) * 4 + (e >> 2), e8 = e & 7, e4 = e & 3, e2 = e & 1;
    pf.a[s_] = stack_at(P.A, b, t)[e];
    const float *Sg = t > 0 ? S.Sigmas_filt + (q - 1) * nn : P.Sigma0 + (int64_t)b * P.Sigma0_sb;
    pf.sg[s_] = Sg[e];
    const float *Sp = S.Sigmas_pred + q * nn;
    pf.sp[s_] = Sp[e];
    pf.spT[s_] = Sp[eT];
    pf.bm[s_] = stack_at(P.Bm, b, t)[e];
    pf.c[s_] = stack_at(P.C, b, t)[e8];
    pf.kv[s_] = S.aux[q * KV_AUX_N4 + e8];
    pf.s4[s_] = S.aux[q * KV_AUX_N4 + 8 + e4];
    pf.mu[s_] = t > 0 ? S.mus_filt[(q - 1) * n + e4] : P.mu0[(int64_t)b * P.mu0_sb + e4];
    pf.mup[s_] = S.mus_pred[q * n + e4];
    pf.u[s_] = P.U[q * m + e4];
    pf.y[s_] = P.Y[q * p + e2];
    pf.mkv[s_] = *mask_addr(P, b, t);
    pf.w_sf[s_] = w[n + e];
    pf.w_sfT[s_] = w[n + eT];
    pf.w_sp[s_] = w[n + nn + n + e];
    pf.w_mf[s_] = w[e4];
    pf.w_mp[s_] = w[n + nn + e4];
    }
  };
  issue(T - 1);
  for (int t = T - 1; t >= 0; --t) {
    const int64_t q = bT + t;
    // ---- commit the prefetched step (with transposed copies), then prefetch the next one ----------------------------------------------
    const FbPf cur = pf;
    KV_PAR(e, nn) {
      const int eT = (e & 3) * 4 + (e >> 2);
      L.A[e] = cur.a[e / KV_LANES];
      L.AT[eT] = cur.a[e / KV_LANES];
      L.Sig[e] = cur.sg[e / KV_LANES];
      L.SigT[eT] = cur.sg[e / KV_LANES];
      L.Sigp[e] = cur.sp[e / KV_LANES];
      L.SpT[eT] = cur.sp[e / KV_LANES];
      L.Sp2[e] = cur.sp[e / KV_LANES] + cur.spT[e / KV_LANES];
      L.Bm[e] = cur.bm[e / KV_LANES];
      if (e < 8) {
        L.C[e] = cur.c[e / KV_LANES];
        L.K[e] = cur.kv[e / KV_LANES];                                     // K[i,c] unmasked, e = i*2 + c
        L.KT[(e & 1) * 4 + (e >> 1)] = cur.kv[e / KV_LANES];
      }
      if (e < 4) {
        L.S[e] = cur.s4[e / KV_LANES];
        L.mu[e] = cur.mu[e / KV_LANES];
        L.mup[e] = cur.mup[e / KV_LANES];
        L.u[e] = cur.u[e / KV_LANES];
        L.gmuT[e] = L.gmu[e] + cur.w_mf[e / KV_LANES];    // total adjoint of mu_f[t] (gmu was written in P7 of the previous iteration)
        L.gmp[e] = cur.w_mp[e / KV_LANES];                // handed-off adjoint of mu_p[t], completed in P6
      }
      if (e < 2) L.y[e] = cur.y[e / KV_LANES];
    }
    KV_LANE0 { L.mk[0] = P.mask ? cur.mkv[0] : 1.0f; }
    if (t > 0) issue(t - 1);
    KV_SYNC();
    const float mk = L.mk[0];
    // ---- P1: totals of the incoming adjoints, G = sym(gSig), IKC (+transpose), r, gr -------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3, eT = j * 4 + i;
      L.G[e] = 0.5f * ((L.gSig[e] + cur.w_sf[e / KV_LANES]) + (L.gSig[eT] + cur.w_sfT[e / KV_LANES]));
      float acc = 0.f;
      acc = fmaf(mk * L.K[i * 2], L.C[j], acc);
      acc = fmaf(mk * L.K[i * 2 + 1], L.C[4 + j], acc);
      const float v = (i == j ? 1.0f : 0.0f) - acc;
      L.IKC[e] = v;
      L.IKCT[eT] = v;
      if (e >= 4 && e < 6) {
        const int c = e - 4;
        const kv4 kc = ld4(L.KT + 4 * c);
        const kv4 gm = ld4(L.gmuT);
        L.r[c] = L.y[c] - fma4(ld4(L.C + 4 * c), ld4(L.mup), 0.f);
        L.gr[c] = fma4(kv4{mk * kc.x, mk * kc.y, mk * kc.z, mk * kc.w}, gm, 0.f);
      }
    }
    KV_SYNC();
    // ---- P2: X1 = G IKC (row i, column j per lane) -> gIKC (+transpose), gSp0 ; GK = G K ------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3, eT = j * 4 + i;
      const M4 Gm = ldm4(L.G), IK = ldm4(L.IKC);
      const kv4 gi_row = ld4(L.G + 4 * i);
      const kv4 x_row = vecmat(gi_row, IK);                                   // X1[i,:]
      const kv4 icj = ld4(L.IKCT + 4 * j);                                    // IKC[:,j]
      const kv4 x_col = kv4{fma4(Gm.r0, icj, 0.f), fma4(Gm.r1, icj, 0.f), fma4(Gm.r2, icj, 0.f), fma4(Gm.r3, icj, 0.f)};
      const float gi = fma4(x_row, ld4(L.Sp2 + 4 * j), 0.f);
      L.gIKC[e] = gi;
      L.gIKCT[eT] = gi;
      L.gSp[e] = fma4(ld4(L.IKCT + 4 * i), x_col, 0.f) + cur.w_sp[e / KV_LANES];
      if (j < p) {
        const kv4 kc = ld4(L.KT + 4 * j);
        L.GK[i * 2 + j] = fma4(gi_row, kv4{mk * kc.x, mk * kc.y, mk * kc.z, mk * kc.w}, 0.f);
      }
    }
    KV_SYNC();
    // ---- P3: gK, gC1 ------------------------------------------------------------------------------------------------------------------------
    KV_PAR(e, n * p) {
      const int i = e >> 1, c = e & 1;
      float acc = 0.f;
      for (int k = 0; k < p; ++k) acc = fmaf(L.GK[i * 2 + k], L.R[c * 2 + k] + L.R[k * 2 + c], acc);
      const kv4 gik = ld4(L.gIKC + 4 * i), cc = ld4(L.C + 4 * c);
      acc = fmaf(-gik.x, cc.x, acc); acc = fmaf(-gik.y, cc.y, acc); acc = fmaf(-gik.z, cc.z, acc); acc = fmaf(-gik.w, cc.w, acc);
      acc = fmaf(L.gmuT[i], L.r[c], acc);
      L.gK[e] = acc;
      const int c2 = e >> 2, j2 = e & 3;   // the same 8 lanes also produce gC1[c2,j2] = -(K^T gIKC)
      const kv4 kc = ld4(L.KT + 4 * c2);
      L.gC1[e] = -fma4(kv4{mk * kc.x, mk * kc.y, mk * kc.z, mk * kc.w}, ld4(L.gIKCT + 4 * j2), 0.f);
    }
    KV_SYNC();
    // ---- P4: Z = solve(S^T, mask gK^T), one column per lane -----------------------------------------------------------------------------------
    KV_PAR(j, n) {
      const kv4 s = ld4(L.S);
      const Sol2 z = solve2(s.x, s.z, s.y, s.w, mk * L.gK[j * 2], mk * L.gK[j * 2 + 1]);
      L.Z[j] = z.x0;
      L.Z[4 + j] = z.x1;
    }
    KV_SYNC();
    // ---- P5: gS0 = sym(-Z Kt^T), Kt = unmasked K^T --------------------------------------------------------------------------------------------------
    KV_PAR(e, p * p) {
      const int a = e >> 1, c = e & 1;
      const float s1 = fma4(ld4(L.Z + 4 * a), ld4(L.KT + 4 * c), 0.f), s2 = fma4(ld4(L.Z + 4 * c), ld4(L.KT + 4 * a), 0.f);
      L.gS0[e] = -0.5f * (s1 + s2);
    }
    KV_SYNC();
    // ---- P6: gSp (final, + transpose) ; gC ; gmp ; gY ------------------------------------------------------------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3, eT = j * 4 + i;
      float acc = L.gSp[e];
      for (int k = 0; k < p; ++k) {
        float gcp = 0.f;  // gCP[k,j] = (gS0 C)[k,j]
        for (int c = 0; c < p; ++c) gcp = fmaf(L.gS0[k * 2 + c], L.C[c * 4 + j], gcp);
        acc = fmaf(L.Z[k * 4 + i], L.C[k * 4 + j], acc);
        acc = fmaf(L.C[k * 4 + i], gcp, acc);
      }
      L.gSpF[e] = acc;
      L.gSpFT[eT] = acc;
      if (G.gQ.ptr) gstack_at(G.gQ, b, t)[e] = acc;
      if (e < 8) {   // lanes 0..7 also produce gC[c,jj]
        const int c = e >> 2, jj = e & 3;
        const kv4 c0 = ld4(L.C), c1 = ld4(L.C + 4);
        const kv4 sptj = ld4(L.SpT + 4 * jj);          // Sig_p[:,jj]
        const float g0 = L.gS0[c * 2], g1 = L.gS0[c * 2 + 1];
        const kv4 gcp = kv4{fmaf(g1, c1.x, g0 * c0.x), fmaf(g1, c1.y, g0 * c0.y), fmaf(g1, c1.z, g0 * c0.z), fmaf(g1, c1.w, g0 * c0.w)};
        float a2 = L.gC1[e];
        a2 = fma4(ld4(L.Z + 4 * c), sptj, a2);                   // (Z Sig_p)[c,jj]
        a2 = fma4(gcp, ld4(L.Sigp + 4 * jj), a2);                // (gCP Sig_p^T)[c,jj]
        a2 = fmaf(g0, fma4(c0, sptj, 0.f), a2);                  // gS0[c,0] (C Sig_p)[0,jj]
        a2 = fmaf(g1, fma4(c1, sptj, 0.f), a2);
        a2 = fmaf(-L.gr[c], L.mup[jj], a2);
        gstack_at(G.gC, b, t)[e] = a2;
      } else if (e < 12) {
        const int ii = e - 8;
        float a3 = L.gmuT[ii] + L.gmp[ii];   // gmp holds the handed-off adjoint of mu_p until here (same lane rewrites it)
        a3 = fmaf(-L.C[ii], L.gr[0], a3);
        a3 = fmaf(-L.C[4 + ii], L.gr[1], a3);
        L.gmp[ii] = a3;
      } else if (e < 14) {
        G.gY[q * p + (e - 12)] = L.gr[e - 12];
      }
    }
    KV_SYNC();
    // ---- P7: gA, carried adjoints of step t-1, gB, gU --------------------------------------------------------------------------------------------------------
    KV_PAR(e, nn) {
      const int i = e >> 2, j = e & 3;
      const M4 Am = ldm4(L.A), gS = ldm4(L.gSpF);
      const kv4 sig_colj = ld4(L.SigT + 4 * j), sig_rowj = ld4(L.Sig + 4 * j);
      const kv4 as_col = kv4{fma4(Am.r0, sig_colj, 0.f), fma4(Am.r1, sig_colj, 0.f), fma4(Am.r2, sig_colj, 0.f), fma4(Am.r3, sig_colj, 0.f)};  // (A Sig)[:,j]
      const kv4 gas_row = vecmat(ld4(L.gSpF + 4 * i), Am);                                                     // (gSp A)[i,:]
      const kv4 atj = ld4(L.AT + 4 * j);
      const kv4 gas_col = kv4{fma4(gS.r0, atj, 0.f), fma4(gS.r1, atj, 0.f), fma4(gS.r2, atj, 0.f), fma4(gS.r3, atj, 0.f)};                     // (gSp A)[:,j]
      float acc = gstack_at(G.gA, b, t)[e];
      acc = fma4(ld4(L.gSpFT + 4 * i), as_col, acc);
      acc = fma4(gas_row, sig_rowj, acc);
      acc = fmaf(L.gmp[i], L.mu[j], acc);
      gstack_at(G.gA, b, t)[e] = acc;
      L.gSig[e] = fma4(ld4(L.AT + 4 * i), gas_col, 0.f);   // gSig was last read in P1
      gstack_at(G.gB, b, t)[e] = L.gmp[i] * L.u[j];        // m == 4
      if (j == 0) L.gmu[i] = fma4(ld4(L.AT + 4 * i), ld4(L.gmp), 0.f);   // gmu was last read in P1
      if (j == 1 && G.gU) {
        const kv4 g4 = ld4(L.gmp);
        G.gU[q * m + i] = fmaf(L.Bm[12 + i], g4.w, fmaf(L.Bm[8 + i], g4.z, fmaf(L.Bm[4 + i], g4.y, L.Bm[i] * g4.x)));
      }
    }
    KV_SYNC();
  }
  if (G.g_mu0) copy_out(G.g_mu0 + (int64_t)b * n, L.gmu, n);
  if (G.g_Sigma0) copy_out(G.g_Sigma0 + (int64_t)b * nn, L.gSig, nn);
}

}  // namespace kvae
