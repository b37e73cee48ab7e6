// lgssm_m4.h — filter + RTS smoother and their adjoints for (n, m, p) = (4, 4, 2) in the quad layout of lgssm_q4.h (sixteen
// sequences per wavefront, lane i of a quad owns row i of every 4x4 matrix) with the 4x4 products on the MATRIX CORES:
// v_mfma_f32_4x4x1_16B_f32 multiplies sixteen independent 4x4 blocks - one per quad - in one instruction.
//
// Why (profiles/r03_q4_fwd_c4_sq.txt, tools/experiments/mfma4_probe.hip): a wavefront of the DPP version of these sweeps issued
// ~235 vector instructions per filter step and ~370 per smoother step, of which the products (16 DPP-folded FMAs each) and the
// transposes (~33 DPP moves and selects each) were the bulk, and with 16 wavefronts on the chip at configs[1] the length of that
// one instruction stream IS the time.  Measured on gfx950: a 4x4x4 product whose result feeds the next product costs 144 clock
// ticks as 16 v_fmac_f32_dpp and 68 as four dependent 4x4x1 MFMAs.
//
// The algebra (R(X) = "rows of X on the lanes of the quad", the register file's view of a matrix):
//     P(X, Y, C) := four MFMAs with A-operand R(X).c[k], B-operand R(Y).c[k], accumulator R(C)   =   R(Y X^T + C)
// (lane j of the quad ends up with row j of Y X^T: D[r][j] = sum_k X[r][k] Y[j][k] sits in register r of lane j).  So
//     A B^T   = P(B, A)                       - the products of the Joseph update and of the smoother are all of this form
//     A S     = P(S, A)      for an EXACTLY symmetric S (the filtered / smoothed covariances, which are stored symmetrised)
//     (Z)^T   = the same four MFMAs with the operands exchanged: (Y X^T + C)^T = X Y^T + C^T = P(Y, X, C^T), bit for bit (the
//               same products summed in the same order) - so no transpose is ever executed: Sigma_p^T, F0^T and Fm^T are
//               computed next to Sigma_p, F0 and Fm (symmetrise = add and halve), a matrix that is in memory anyway is read by
//               columns, and J = (J^T)^T is P(J^T, I) - a product with the identity, exact.
//     y x^T   = ONE MFMA on two vectors held one entry per lane (outer1): I - K C, K R K^T and the rank-one terms of the adjoint
//               need no quad broadcast.
// Same equations, same summation order per element (k ascending, accumulator first) as the reference
// (kalman_filter.py:31-104, 204-237; adjoint: lgssm_bwd.h); same outputs, aux record (K unmasked | S | J) and ws hand-off record
// as lgssm_n4.h.  Everything except the pivoted 4x4 solve is compiler-visible code (builtins, no inline assembly): the
// MFMA -> VALU / VALU -> MFMA hazards are the compiler's to handle; the pivoted solve (inline-asm DPP FMAs of lgssm_q4.h, the
// rare path) is fenced by explicit wait states on both sides.  tools/m4_selftest.hip checks every primitive against plain loops.
#pragma once
#include <stdlib.h>

#include "lgssm_q4.h"

#if (!defined(KVAE_HOSTSIM) || defined(KVAE_WAVE_EMU)) && !defined(KV_TPP)   // KVAE_WAVE_EMU: tests/hostsim/wave_emu.h
namespace kvae {
namespace m4 {

using q4::f2;
using q4::f4;
using q4::Mat;

// R(Y X^T + C) from R(X), R(Y), R(C)
__device__ __forceinline__ Mat P(const Mat &X, const Mat &Y, const Mat &C) {
  // (two accumulators of two k-steps each plus an add - a dependent chain of 2 MFMAs instead of 4 - was measured: 1-2 % slower)
  f4 c = f4{C.c[0], C.c[1], C.c[2], C.c[3]};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(X.c[0], Y.c[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(X.c[1], Y.c[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(X.c[2], Y.c[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(X.c[3], Y.c[3], c, 0, 0, 0);
  return Mat{{c[0], c[1], c[2], c[3]}};
}
__device__ __forceinline__ Mat P(const Mat &X, const Mat &Y) { return P(X, Y, q4::zero()); }
// rank-one terms on the same instruction: R(C + y x^T) for two vectors held one entry per lane (lane j gets y_j x_r in register
// r) - outer products without broadcasting either vector over the quad
__device__ __forceinline__ Mat outer1(float x, float y, const Mat &C) {
  f4 c = f4{C.c[0], C.c[1], C.c[2], C.c[3]};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(x, y, c, 0, 0, 0);
  return Mat{{c[0], c[1], c[2], c[3]}};
}
__device__ __forceinline__ Mat outer2(float x0, float y0, float x1, float y1, const Mat &C) {   // C + y0 x0^T + y1 x1^T
  return outer1(x1, y1, outer1(x0, y0, C));
}
// every register of M has landed / may be overwritten: wait states around the inline-asm code of q4::solve, whose reads and
// writes the compiler's hazard recogniser cannot see (XDL write -> VALU read of a 2-pass MFMA: 5; VALU write -> XDL read: 2)
#if defined(KVAE_WAVE_EMU)
__device__ __forceinline__ void settle(Mat &) {}
#else
__device__ __forceinline__ void settle(Mat &M) { asm volatile("s_nop 7" : "+v"(M.c[0]), "+v"(M.c[1]), "+v"(M.c[2]), "+v"(M.c[3])); }
#endif

// all four entries of a quad's vector on every lane of the quad
struct Vec4 { float c[4]; };
__device__ __forceinline__ Vec4 spread(float v) { return Vec4{{q4::qb<0>(v), q4::qb<1>(v), q4::qb<2>(v), q4::qb<3>(v)}}; }
__device__ __forceinline__ float dot(const Mat &Mrow, const Vec4 &v, float acc) {      // acc + sum_c M[i][c] v[c]
  return fmaf(Mrow.c[3], v.c[3], fmaf(Mrow.c[2], v.c[2], fmaf(Mrow.c[1], v.c[1], fmaf(Mrow.c[0], v.c[0], acc))));
}
__device__ __forceinline__ Mat load_cols(const float *X, int i) {   // R(X^T): lane i takes column i of a row-major 4x4
  return Mat{{X[i], X[4 + i], X[8 + i], X[12 + i]}};
}
__device__ __forceinline__ Mat half_sum(const Mat &A, const Mat &B) {
  Mat S;
#pragma unroll
  for (int c = 0; c < 4; ++c) S.c[c] = 0.5f * (A.c[c] + B.c[c]);
  return S;
}

// ---- 4x4 solve  M X = W  (rows of M and of W on the lanes) in natural order ---------------------------------------------------
// The reference calls torch.linalg.solve (getrf, partial pivoting).  Both systems of this path have a predicted covariance as
// their matrix - symmetric positive definite, for which elimination without exchanges is backward stable - so the fast path
// eliminates in natural order (as lgssm_n16.h does): straight-line, compiler-visible code the scheduler can overlap with the
// products around it.  It watches the pivots; one that is not positive and finite sets `bad`, and the caller repeats the solve
// with q4::solve (getrf's pivot sequence) at the end of the step.  Without an exchange the two produce the same bits.
template <int K>
__device__ __forceinline__ void gj_natural(Mat &m, Mat &x, float &my_rinv, bool &bad, int i) {
  const float piv = q4::qb<K>(m.c[K]);
  bad |= !(piv > 0.0f && piv < INFINITY);
  const float rinv = q4::frcp(piv);
  const float f = i == K ? 0.0f : -(m.c[K] * rinv);
  my_rinv = i == K ? rinv : my_rinv;
#pragma unroll
  for (int c = K + 1; c < 4; ++c) m.c[c] = fmaf(q4::qb<K>(m.c[c]), f, m.c[c]);
#pragma unroll
  for (int c = 0; c < 4; ++c) x.c[c] = fmaf(q4::qb<K>(x.c[c]), f, x.c[c]);
  if constexpr (K + 1 < 4) gj_natural<K + 1>(m, x, my_rinv, bad, i);
}
__device__ __forceinline__ Mat solve_natural(Mat m, Mat x, int i, bool &bad) {
  float my_rinv = 0.0f;
  bad = false;
  gj_natural<0>(m, x, my_rinv, bad, i);
#pragma unroll
  for (int c = 0; c < 4; ++c) x.c[c] *= my_rinv;
  return x;
}
__device__ __forceinline__ Mat solve_pivoted(Mat m, Mat x, int i, int lane) {   // the rare path: inline-asm code, fenced
  settle(m);
  settle(x);
  Mat r = q4::solve(m, x, i, lane);
  settle(r);
  return r;
}

struct StepIn {
  Mat A, Bm, Q, Qt;
  Vec4 C0, C1;          // the two emission rows, whole, on every lane
  float Cl0, Cl1, u, mk;
  f2 y;
};
__device__ __forceinline__ void load_step(const q4::StepPtr &p, int i, StepIn &s) {
  s.A = q4::load_rows(p.A, i), s.Bm = q4::load_rows(p.Bm, i), s.Q = q4::load_rows(p.Q, i);
  s.Qt = load_cols(p.Q, i);
  const f4 c0 = *reinterpret_cast<const f4 *>(p.C), c1 = *reinterpret_cast<const f4 *>(p.C + 4);
  s.C0 = Vec4{{c0[0], c0[1], c0[2], c0[3]}}, s.C1 = Vec4{{c1[0], c1[1], c1[2], c1[3]}};
  s.Cl0 = p.C[i], s.Cl1 = p.C[4 + i];
  s.u = p.U[i];
  s.y = *reinterpret_cast<const f2 *>(p.Y);
  s.mk = *p.mk;   // raw; the NULL-mask select happens at the point of use (see lgssm_n16.h)
}

// HOIST: the smoother gain J_{t-1} = Sig_f[t-1] A_t^T Sig_p[t]^{-1} (kalman_filter.py:229) does not depend on anything the
// backward recursion carries, and its operands - A_t Sig_f[t-1], Sig_p[t]^T - are in registers in filter step t.  It is
// computed there, off the filter's dependent chain (the scheduler fills the chain's latency with it), and left in the aux
// record (or, for a caller without one, in the Sigmas_smooth slot of t-1, which the smoother overwrites after reading it).
template <bool AUX>
__device__ __forceinline__ float *gain_slot(const kvae_lgssm_states &S, int64_t q) {
  return AUX ? S.aux + q * KV_AUX_N4 + 12 : S.Sigmas_smooth + q * 16;
}

template <bool AUX, bool HOIST>
__device__ __forceinline__ void filter_sweep(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S, int b, int i, int lane) {
  const int T = P_.T;
  const int64_t bT = (int64_t)b * T;
  Mat Sig = load_cols(P_.Sigma0 + (int64_t)b * P_.Sigma0_sb, i);   // Sigma0^T, so that P(Sig, A) = A Sigma0 for any Sigma0
  float mu = P_.mu0[(int64_t)b * P_.mu0_sb + i];
  const float R00 = P_.R[0], R01 = P_.R[1], R10 = P_.R[2], R11 = P_.R[3];
  const Mat I4 = q4::eye(i);
  q4::StepPtr ptr;
  ptr.init(P_, b, 0);
  // Operands TWO steps ahead: a chain step is ~0.5 us, less than an HBM miss under load (SQ counters of the one-step version,
  // profiles/r03_m4_{fwd,bwd}_sq.txt: 18-23 % of every sweep in s_waitcnt).  Three operand sets that rotate by NAME - the caller
  // is unrolled three times - because a register copy of a prefetched value waits for its load.
  StepIn s0, s1, s2;
  load_step(ptr, i, s0);
  if (T > 1) ptr.step(1);
  load_step(ptr, i, s1);
  s2 = s1;
  KV_Q4_DRAIN();
  // one step: operands in `s`, those of step t + 2 fetched into `far`
  auto step = [&](int t, const StepIn &s, StepIn &far) {
    if (t + 2 < T) ptr.step(1);
    load_step(ptr, i, far);                                  // unconditional prefetch, pinned above this step's stores
    KV_Q4_FENCE();
    const int64_t q = bT + t;
    // predict (kalman_filter.py:65-67)
    const float mup = dot(s.Bm, spread(s.u), dot(s.A, spread(mu), 0.0f));
    const Mat AS = P(Sig, s.A);                              // A Sig   (Sig symmetric)
    const Mat Sigp = P(s.A, AS, s.Q);                        // (A Sig) A^T + Q
    const Mat Sigpt = P(AS, s.A, s.Qt);                      // its transpose, bit for bit
    bool bad = false;
    if constexpr (HOIST) {                                   // Sig_p[t]^T J^T = A_t Sig_f[t-1]; step 0 writes a dummy into slot 0
      const Mat X = solve_natural(Sigpt, AS, i, bad);
      q4::store_rows(gain_slot<AUX>(S, t > 0 ? q - 1 : q), P(X, I4), i);
      bad = bad && t > 0;
    }
    q4::store_rows(S.Sigmas_pred + q * 16, Sigp, i);
    S.mus_pred[q * 4 + i] = mup;
    // innovation (:73-90): PCT = Sigp C^T (lane i: PCT[i][c]), S = sym(C PCT + R), r = y - C mup
    const float pct0 = dot(Sigp, s.C0, 0.0f), pct1 = dot(Sigp, s.C1, 0.0f);
    const float a00 = q4::qsum(s.Cl0 * pct0) + R00, a01 = q4::qsum(s.Cl0 * pct1) + R01;
    const float a10 = q4::qsum(s.Cl1 * pct0) + R10, a11 = q4::qsum(s.Cl1 * pct1) + R11;
    const float s00 = 0.5f * (a00 + a00), s01 = 0.5f * (a01 + a10), s11 = 0.5f * (a11 + a11);
    const float r0 = s.y[0] - q4::qsum(s.Cl0 * mup), r1 = s.y[1] - q4::qsum(s.Cl1 * mup);
    const q4::Inv2 F = q4::factor2(s00, s01, s11);
    float ku0, ku1;
    q4::solve2(F, pct0, pct1, ku0, ku1);                     // unmasked gain K[i][:]
    if constexpr (AUX) {
      float *ax = S.aux + q * KV_AUX_N4;
      *reinterpret_cast<f2 *>(ax + 2 * i) = f2{ku0, ku1};
      *reinterpret_cast<f4 *>(ax + 8) = f4{s00, s01, s01, s11};
    }
    const float mk = P_.mask ? s.mk : 1.0f;
    const float k0 = mk * ku0, k1 = mk * ku1;                // :92
    const float muf = mup + k0 * r0 + k1 * r1;               // :96
    S.mus_filt[q * 4 + i] = muf;
    // Joseph update (:97-101): M = I - K C ; (M Sigp) M^T + (K R) K^T, and its transpose next to it
    const float kr0 = k0 * R00 + k1 * R10, kr1 = k0 * R01 + k1 * R11;
    const Mat M = outer2(s.Cl0, -k0, s.Cl1, -k1, I4);        // I - k0 C0^T - k1 C1^T
    const Mat KRK = outer2(k0, kr0, k1, kr1, q4::zero());    // (K R) K^T
    const Mat KRKt = outer2(kr0, k0, kr1, k1, q4::zero());   // K (K R)^T
    const Mat T1 = P(Sigpt, M);                              // M Sigp
    const Mat F0 = P(M, T1, KRK);                            // (M Sigp) M^T + K R K^T
    const Mat F0t = P(T1, M, KRKt);                          // its transpose, bit for bit
    Sig = half_sum(F0, F0t);                                 // :101, exactly symmetric
    q4::store_rows(S.Sigmas_filt + q * 16, Sig, i);
    mu = muf;
    if constexpr (HOIST) {
      if (__any(bad)) q4::store_rows(gain_slot<AUX>(S, q - 1), P(solve_pivoted(Sigpt, AS, i, lane), I4), i);
    }
  };
  int t = 0;
  for (; t + 2 < T; t += 3) {
    step(t, s0, s2);
    step(t + 1, s1, s0);
    step(t + 2, s2, s1);
  }
  if (t < T) step(t, s0, s2);
  if (t + 1 < T) step(t + 1, s1, s0);
}

template <bool AUX, bool HAVE_J>
__device__ __forceinline__ void rts_sweep(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S, int b, int i, int lane) {
  const int T = P_.T;
  const int64_t bT = (int64_t)b * T;
  Mat SigS = q4::load_rows(S.Sigmas_filt + (bT + T - 1) * 16, i);
  float mus = S.mus_filt[(bT + T - 1) * 4 + i];
  q4::store_rows(S.Sigmas_smooth + (bT + T - 1) * 16, SigS, i);
  S.mus_smooth[(bT + T - 1) * 4 + i] = mus;
  struct In { Mat Sf, Sp, Spt, A; float muf, mup; } s0, s1, s2;   // A: the gain J_t itself when the filter sweep left it (HAVE_J)
  int64_t q = bT + (T >= 2 ? T - 2 : 0);
  const float *pA = stack_at(P_.A, b, T >= 2 ? T - 1 : 0);
  const int64_t sA = P_.A.st;
  auto load = [&](In &o) {
    o.Sf = q4::load_rows(S.Sigmas_filt + q * 16, i);
    o.Sp = q4::load_rows(S.Sigmas_pred + (q + 1) * 16, i);
    o.Spt = load_cols(S.Sigmas_pred + (q + 1) * 16, i);
    o.A = q4::load_rows(HAVE_J ? gain_slot<AUX>(S, q) : pA, i);
    o.muf = S.mus_filt[q * 4 + i];
    o.mup = S.mus_pred[(q + 1) * 4 + i];
  };
  if (T >= 2) load(s0);
  s1 = s0;
  if (T >= 3) {
    q -= 1, pA -= sA;
    load(s1);
  }
  s2 = s1;
  KV_Q4_DRAIN();
  const Mat I4 = q4::eye(i);
  auto step = [&](int t, const In &s, In &far) {              // operands two steps ahead, rotated by name: see filter_sweep
    const int64_t qt = bT + t;
    if (t >= 2) q -= 1, pA -= sA;
    load(far);
    KV_Q4_FENCE();
    Mat J = s.A;
    if constexpr (!HAVE_J) {
      // J = Sig_f A^T Sigp^{-1}  <=>  Sigp^T J^T = A Sig_f  (kalman_filter.py:229)
      const Mat W = P(s.Sf, s.A);                            // A Sig_f   (Sig_f symmetric)
      bool bad;
      Mat X = solve_natural(s.Spt, W, i, bad);               // J^T, rows on lanes
      if (__any(bad)) X = solve_pivoted(s.Spt, W, i, lane);
      J = P(X, I4);                                          // (J^T)^T: a product with the identity, exact
      if constexpr (AUX) q4::store_rows(S.aux + qt * KV_AUX_N4 + 12, J, i);
    }
    const Mat D = q4::sub(SigS, s.Sp), Dt = q4::sub(SigS, s.Spt);   // D and D^T (SigS symmetric)
    const Mat G = P(Dt, J);                                  // J D
    mus = dot(J, spread(mus - s.mup), s.muf);                // :232
    const Mat Fm = P(J, G, s.Sf);                            // Sig_f + (J D) J^T   (:234)
    const Mat Fmt = P(G, J, s.Sf);                           // its transpose (Sig_f symmetric)
    SigS = half_sum(Fm, Fmt);
    q4::store_rows(S.Sigmas_smooth + qt * 16, SigS, i);
    S.mus_smooth[qt * 4 + i] = mus;
  };
  int t = T - 2;
  for (; t >= 2; t -= 3) {
    step(t, s0, s2);
    step(t - 1, s1, s0);
    step(t - 2, s2, s1);
  }
  if (t >= 0) step(t, s0, s2);
  if (t >= 1) step(t - 1, s1, s0);
}


// =====================================================================================================================
// backward: the adjoint sweeps of lgssm_q4.h (same equations, same ws hand-off record, same saved gains K | S | J), products on
// the matrix cores.  Every transpose the quad kernels execute (six per smoother step, seven per filter step) is either a column
// load of a matrix that is in memory anyway, a second product with the operands exchanged, or a product with the identity.
// =====================================================================================================================
__device__ __forceinline__ Vec4 load_vec(const float *v) {   // a 4-vector, whole, on every lane
  const f4 x = *reinterpret_cast<const f4 *>(v);
  return Vec4{{x[0], x[1], x[2], x[3]}};
}

template <bool HAS_FP>
__device__ __forceinline__ void rts_bwd_sweep(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                                              const kvae_lgssm_input_grads &G, float *ws, int b, int i, int lane) {
  constexpr int WS_REC = q4::WS_REC;
  const int T = P_.T;
  const int64_t bT = (int64_t)b * T;
  float *w = ws + bT * WS_REC;
  float gsm = U.mus_smooth[bT * 4 + i];
  Mat gsS = q4::load_rows(U.Sigmas_smooth + bT * 16, i), gsSt = load_cols(U.Sigmas_smooth + bT * 16, i);
  w[4 + 16 + i] = HAS_FP ? U.mus_pred[bT * 4 + i] : 0.0f;
  q4::store_rows(w + 4 + 16 + 4, HAS_FP ? q4::load_rows(U.Sigmas_pred + bT * 16, i) : q4::zero(), i);
  q4::store_rows(gstack_at(G.gA, b, 0), q4::zero(), i);
  struct In {
    Mat Sf, Sp, Spt, Ss, At, Jt, uSs, uSst, uSf, uSp;
    float mup, mus, uMs, uMf, uMp;
  } s, nx;
  int64_t q = bT;
  const float *pA = stack_at(P_.A, b, T >= 2 ? 1 : 0);
  const int64_t sA = P_.A.st;
  auto load = [&](In &o) {
    o.Sf = q4::load_rows(S.Sigmas_filt + q * 16, i);
    o.Sp = q4::load_rows(S.Sigmas_pred + (q + 1) * 16, i);
    o.Spt = load_cols(S.Sigmas_pred + (q + 1) * 16, i);
    o.Ss = q4::load_rows(S.Sigmas_smooth + (q + 1) * 16, i);
    o.At = load_cols(pA, i);
    o.Jt = load_cols(S.aux + q * KV_AUX_N4 + 12, i);
    o.mup = S.mus_pred[(q + 1) * 4 + i];
    o.mus = S.mus_smooth[(q + 1) * 4 + i];
    o.uMs = U.mus_smooth[(q + 1) * 4 + i];
    o.uSs = q4::load_rows(U.Sigmas_smooth + (q + 1) * 16, i);
    o.uSst = load_cols(U.Sigmas_smooth + (q + 1) * 16, i);
    if constexpr (HAS_FP) {
      o.uMf = U.mus_filt[q * 4 + i];
      o.uSf = q4::load_rows(U.Sigmas_filt + q * 16, i);
      o.uMp = U.mus_pred[(q + 1) * 4 + i];
      o.uSp = q4::load_rows(U.Sigmas_pred + (q + 1) * 16, i);
    }
  };
  if (T >= 2) load(s);
  nx = s;
  KV_Q4_DRAIN();
  const Mat I4 = q4::eye(i);
  auto step = [&](int t, const In &s, In &nx) {
    if (t + 2 < T) q += 1, pA += sA;
    load(nx);
    KV_Q4_FENCE();
    float *wt = w + (int64_t)t * WS_REC;
    const Mat gM = half_sum(gsS, gsSt);                               // sym(adjoint of Sig_s[t])
    const Mat D2 = q4::add(q4::sub(s.Ss, s.Sp), q4::sub(s.Ss, s.Spt));   // D + D^T, D = Sig_s[t+1] - Sig_p[t+1]: symmetric
    const float dmu = s.mus - s.mup;
    const Vec4 gsmv = spread(gsm);
    const Mat Y1 = P(s.Jt, gM), Y1t = P(gM, s.Jt);                    // gM J and its transpose
    Mat O;
#pragma unroll
    for (int c = 0; c < 4; ++c) O.c[c] = dmu * gsmv.c[c];
    const Mat gJt = P(Y1, D2, O);                                         // (Y1 (D^T + D) + gsm dmu^T)^T
    const Mat gD = P(Y1t, s.Jt), gDt = P(s.Jt, Y1t);                  // J^T Y1 and its transpose
    const float gdm = dot(s.Jt, gsmv, 0.0f);                          // J^T gsm
    auto finish = [&](const Mat &gR) {                                // everything that hangs off the solve: stores only
      const Mat gRt = P(gR, I4);
      const Mat gWA = P(s.At, gRt);                                   // gR^T A[t+1]
      const Mat gP = P(gR, s.Jt);                                     // J^T gR^T
      const Mat gAs = P(s.Sf, gR);                                    // gR Sig_f: the smoother's share of gA[t+1]
      q4::store_rows(wt + 4, q4::add(q4::add(HAS_FP ? s.uSf : q4::zero(), gM), gWA), i);
      q4::store_rows(wt + WS_REC + 4 + 16 + 4, q4::sub(q4::sub(HAS_FP ? s.uSp : q4::zero(), gD), gP), i);
      q4::store_rows(gstack_at(G.gA, b, t + 1), gAs, i);
    };
    bool bad;
    finish(solve_natural(s.Sp, gJt, i, bad));                         // Sig_p gR = gJ^T
    wt[i] = (HAS_FP ? s.uMf : 0.0f) + gsm;
    wt[WS_REC + 4 + 16 + i] = (HAS_FP ? s.uMp : 0.0f) - gdm;
    gsS = q4::add(s.uSs, gD);
    gsSt = q4::add(s.uSst, gDt);
    gsm = s.uMs + gdm;
    if (__any(bad)) finish(solve_pivoted(s.Sp, gJt, i, lane));
  };
  int t = 0;
  for (; t + 2 < T; t += 2) {
    step(t, s, nx);
    step(t + 1, nx, s);
  }
  if (t + 1 < T) step(t, s, nx);
  float *wl = w + (int64_t)(T - 1) * WS_REC;
  const int64_t ql = bT + T - 1;
  q4::store_rows(wl + 4, q4::add(HAS_FP ? q4::load_rows(U.Sigmas_filt + ql * 16, i) : q4::zero(), gsS), i);
  wl[i] = (HAS_FP ? U.mus_filt[ql * 4 + i] : 0.0f) + gsm;
}

// ---- the smoother's adjoint in two parts, for batches far below the chip's wave slots (kv_m4_split_bwd) ------------------------------
// Of the ~270 instructions of an rts_bwd_sweep step only ~60 are on the chain that step t + 1 waits for: the adjoint of the
// smoothed belief travels gM -> Y1^T -> gD (three products).  Everything else - the solve for the adjoint of the gain, four more
// products, all of the hand-off record - hangs off it and is B (T - 1) independent problems.  rts_bwd_chain runs the chain alone and
// leaves (gM, gsm) of every step in two OUTPUT slots that are written later anyway (gA[t+1]: the smoother's share lands there;
// gU[t]: written by the filter's adjoint); rts_bwd_items turns them into the hand-off records, sixteen (b, t) per wavefront.  Same
// operations on the same operands as rts_bwd_sweep: the two forms produce the same bits.
template <bool HAS_FP>
__device__ __forceinline__ void rts_bwd_chain(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                                              const kvae_lgssm_input_grads &G, float *ws, int b, int i) {
  constexpr int WS_REC = q4::WS_REC;
  const int T = P_.T;
  const int64_t bT = (int64_t)b * T;
  float *w = ws + bT * WS_REC;
  float gsm = U.mus_smooth[bT * 4 + i];
  Mat gsS = q4::load_rows(U.Sigmas_smooth + bT * 16, i), gsSt = load_cols(U.Sigmas_smooth + bT * 16, i);
  w[4 + 16 + i] = HAS_FP ? U.mus_pred[bT * 4 + i] : 0.0f;
  q4::store_rows(w + 4 + 16 + 4, HAS_FP ? q4::load_rows(U.Sigmas_pred + bT * 16, i) : q4::zero(), i);
  q4::store_rows(gstack_at(G.gA, b, 0), q4::zero(), i);
  struct In { Mat Jt, uSs, uSst; float uMs; } s0, s1, s2;
  int64_t q = bT;
  auto load = [&](In &o) {
    o.Jt = load_cols(S.aux + q * KV_AUX_N4 + 12, i);
    o.uMs = U.mus_smooth[(q + 1) * 4 + i];
    o.uSs = q4::load_rows(U.Sigmas_smooth + (q + 1) * 16, i);
    o.uSst = load_cols(U.Sigmas_smooth + (q + 1) * 16, i);
  };
  if (T >= 2) load(s0);
  s1 = s0;
  if (T >= 3) {
    q += 1;
    load(s1);
  }
  s2 = s1;
  KV_Q4_DRAIN();
  auto step = [&](int t, const In &s, In &far) {                      // operands two steps ahead, rotated by name: see filter_sweep
    if (t + 3 < T) q += 1;
    load(far);
    KV_Q4_FENCE();
    const Mat gM = half_sum(gsS, gsSt);                               // sym(adjoint of Sig_s[t])
    q4::store_rows(gstack_at(G.gA, b, t + 1), gM, i);                 // parked for rts_bwd_items
    G.gU[(bT + t) * 4 + i] = gsm;
    const Mat Y1t = P(gM, s.Jt);                                      // (gM J)^T
    const Mat gD = P(Y1t, s.Jt), gDt = P(s.Jt, Y1t);                  // J^T (gM J) and its transpose
    const float gdm = dot(s.Jt, spread(gsm), 0.0f);                   // J^T gsm
    gsS = q4::add(s.uSs, gD);
    gsSt = q4::add(s.uSst, gDt);
    gsm = s.uMs + gdm;
  };
  int t = 0;
  for (; t + 3 < T; t += 3) {
    step(t, s0, s2);
    step(t + 1, s1, s0);
    step(t + 2, s2, s1);
  }
  if (t + 1 < T) step(t, s0, s2);
  if (t + 2 < T) step(t + 1, s1, s0);
  float *wl = w + (int64_t)(T - 1) * WS_REC;
  const int64_t ql = bT + T - 1;
  q4::store_rows(wl + 4, q4::add(HAS_FP ? q4::load_rows(U.Sigmas_filt + ql * 16, i) : q4::zero(), gsS), i);
  wl[i] = (HAS_FP ? U.mus_filt[ql * 4 + i] : 0.0f) + gsm;
}

template <bool HAS_FP>
__device__ __forceinline__ void rts_bwd_items(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                                              const kvae_lgssm_input_grads &G, float *ws) {
  constexpr int WS_REC = q4::WS_REC;
  const int lane = threadIdx.x & 63, i = lane & 3;
  const int64_t items = (int64_t)P_.B * (P_.T - 1);
  int64_t it = (int64_t)blockIdx.x * 16 + (lane >> 2);
  it = it < items ? it : items - 1;      // a ragged last wavefront recomputes the last item
  const int b = (int)(it / (P_.T - 1)), t = (int)(it - (int64_t)b * (P_.T - 1));
  const int64_t q = (int64_t)b * P_.T + t;
  float *wt = ws + q * WS_REC;
  float *gA1 = gstack_at(G.gA, b, t + 1);
  const Mat gM = q4::load_rows(gA1, i);
  const float gsm = G.gU[q * 4 + i];
  const Mat Jt = load_cols(S.aux + q * KV_AUX_N4 + 12, i), Sf = q4::load_rows(S.Sigmas_filt + q * 16, i);
  const Mat Sp = q4::load_rows(S.Sigmas_pred + (q + 1) * 16, i), Spt = load_cols(S.Sigmas_pred + (q + 1) * 16, i);
  const Mat Ss = q4::load_rows(S.Sigmas_smooth + (q + 1) * 16, i), At = load_cols(stack_at(P_.A, b, t + 1), i);
  const float dmu = S.mus_smooth[(q + 1) * 4 + i] - S.mus_pred[(q + 1) * 4 + i];
  const Mat uSf = HAS_FP ? q4::load_rows(U.Sigmas_filt + q * 16, i) : q4::zero();
  const Mat uSp = HAS_FP ? q4::load_rows(U.Sigmas_pred + (q + 1) * 16, i) : q4::zero();
  const float uMf = HAS_FP ? U.mus_filt[q * 4 + i] : 0.0f, uMp = HAS_FP ? U.mus_pred[(q + 1) * 4 + i] : 0.0f;
  const Mat D2 = q4::add(q4::sub(Ss, Sp), q4::sub(Ss, Spt));          // D + D^T, D = Sig_s[t+1] - Sig_p[t+1]: symmetric
  const Vec4 gsmv = spread(gsm);
  const Mat Y1 = P(Jt, gM), Y1t = P(gM, Jt);                          // gM J and its transpose
  Mat O;
#pragma unroll
  for (int c = 0; c < 4; ++c) O.c[c] = dmu * gsmv.c[c];
  const Mat gJt = P(Y1, D2, O);                                       // (Y1 (D^T + D) + gsm dmu^T)^T
  const Mat gD = P(Y1t, Jt);                                          // J^T Y1
  const float gdm = dot(Jt, gsmv, 0.0f);                              // J^T gsm
  bool bad;
  Mat gR = solve_natural(Sp, gJt, i, bad);                            // Sig_p gR = gJ^T
  if (__any(bad)) gR = solve_pivoted(Sp, gJt, i, lane);
  const Mat gRt = P(gR, q4::eye(i));
  const Mat gWA = P(At, gRt);                                         // gR^T A[t+1]
  const Mat gP = P(gR, Jt);                                           // J^T gR^T
  const Mat gAs = P(Sf, gR);                                          // gR Sig_f: the smoother's share of gA[t+1]
  q4::store_rows(wt + 4, q4::add(q4::add(uSf, gM), gWA), i);
  q4::store_rows(wt + WS_REC + 4 + 16 + 4, q4::sub(q4::sub(uSp, gD), gP), i);
  q4::store_rows(gA1, gAs, i);
  wt[i] = uMf + gsm;
  wt[WS_REC + 4 + 16 + i] = uMp - gdm;
}

template <bool HAS_GQ>
__device__ __forceinline__ void filter_bwd_sweep(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S,
                                                 const kvae_lgssm_input_grads &G, const float *ws, int b, int i, int lane) {
  constexpr int WS_REC = q4::WS_REC;
  const int T = P_.T;
  const int64_t bT = (int64_t)b * T;
  const float R00 = P_.R[0], R01 = P_.R[1], R10 = P_.R[2], R11 = P_.R[3];
  const Mat I4 = q4::eye(i);
  struct In {
    Mat A, At, Bt, Sig, Sigt, Sp, Spt, wSf, wSft, wSp, gAs;
    Vec4 C0, C1, u, mu;
    float Cl0, Cl1, mk, mup, wmf, wmp;
    f2 y, ku;
    f4 Sv;
  } s, nx;
  q4::StepPtr ptr;
  ptr.init(P_, b, T - 1);
  int t_ld = T - 1;
  auto load = [&](In &o) {
    const int64_t q = bT + t_ld;
    o.A = q4::load_rows(ptr.A, i), o.At = load_cols(ptr.A, i), o.Bt = load_cols(ptr.Bm, i);
    o.C0 = load_vec(ptr.C), o.C1 = load_vec(ptr.C + 4);
    o.Cl0 = ptr.C[i], o.Cl1 = ptr.C[4 + i];
    o.u = load_vec(ptr.U);
    o.y = *reinterpret_cast<const f2 *>(ptr.Y);
    o.mk = *ptr.mk;
    const float *pS = t_ld > 0 ? S.Sigmas_filt + (q - 1) * 16 : P_.Sigma0 + (int64_t)b * P_.Sigma0_sb;
    const float *pm = t_ld > 0 ? S.mus_filt + (q - 1) * 4 : P_.mu0 + (int64_t)b * P_.mu0_sb;
    o.Sig = q4::load_rows(pS, i), o.Sigt = load_cols(pS, i);
    o.mu = load_vec(pm);
    o.Sp = q4::load_rows(S.Sigmas_pred + q * 16, i), o.Spt = load_cols(S.Sigmas_pred + q * 16, i);
    o.mup = S.mus_pred[q * 4 + i];
    const float *ax = S.aux + q * KV_AUX_N4;
    o.ku = *reinterpret_cast<const f2 *>(ax + 2 * i);
    o.Sv = *reinterpret_cast<const f4 *>(ax + 8);
    const float *w = ws + q * WS_REC;
    o.wmf = w[i];
    o.wSf = q4::load_rows(w + 4, i), o.wSft = load_cols(w + 4, i);
    o.wmp = w[4 + 16 + i];
    o.wSp = q4::load_rows(w + 4 + 16 + 4, i);
    o.gAs = q4::load_rows(gstack_at(G.gA, b, t_ld), i);
  };
  load(s);
  nx = s;
  KV_Q4_DRAIN();
  float gmu = 0.0f;
  Mat gSig = q4::zero(), gSigt = q4::zero();
  auto step = [&](int t, const In &s, In &nx) {
    if (t >= 1) t_ld = t - 1, ptr.step(-1);
    load(nx);
    KV_Q4_FENCE();
    const int64_t q = bT + t;
    const float mk = P_.mask ? s.mk : 1.0f;
    gmu += s.wmf;
    gSig = q4::add(gSig, s.wSf), gSigt = q4::add(gSigt, s.wSft);
    const Mat Gm = half_sum(gSig, gSigt);
    const float k0 = mk * s.ku[0], k1 = mk * s.ku[1];
    const Vec4 k0v = spread(k0), k1v = spread(k1);
    const Mat Mt = outer2(-k0, s.Cl0, -k1, s.Cl1, I4);                // (I - K C)^T = I - C0 k0^T - C1 k1^T
    const float gr0 = q4::qsum(k0 * gmu), gr1 = q4::qsum(k1 * gmu);   // gr = K^T gmu
    const float r0 = s.y[0] - q4::qsum(s.Cl0 * s.mup), r1 = s.y[1] - q4::qsum(s.Cl1 * s.mup);
    const Mat Sp2 = q4::add(s.Sp, s.Spt);                             // symmetric
    const Mat X1 = P(Mt, Gm), X1t = P(Gm, Mt);                        // G (I - K C) and its transpose
    const Mat gIKC = P(Sp2, X1), gIKCt = P(X1, Sp2);                  // X1 (Sig_p^T + Sig_p) and its transpose
    Mat gSp = P(X1t, Mt, s.wSp);                                      // (I - K C)^T X1 + handed-off
    // gK = G K (R^T + R) - gIKC C^T + gmu r^T   (lane i: row i)
    const float GK0 = dot(Gm, k0v, 0.0f), GK1 = dot(Gm, k1v, 0.0f);
    const float gK0 = GK0 * (R00 + R00) + GK1 * (R01 + R10) - dot(gIKC, s.C0, 0.0f) + gmu * r0;
    const float gK1 = GK0 * (R10 + R01) + GK1 * (R11 + R11) - dot(gIKC, s.C1, 0.0f) + gmu * r1;
    // Z = S^{-T} (mask gK^T): column i on lane i
    const q4::Inv2 F = q4::factor2(s.Sv[0], s.Sv[1], s.Sv[3]);
    float z0, z1;
    q4::solve2(F, mk * gK0, mk * gK1, z0, z1);
    const float zk00 = q4::qsum(z0 * s.ku[0]), zk01 = q4::qsum(z0 * s.ku[1]), zk10 = q4::qsum(z1 * s.ku[0]),
                zk11 = q4::qsum(z1 * s.ku[1]);
    const float h00 = -0.5f * (zk00 + zk00), h01 = -0.5f * (zk01 + zk10), h11 = -0.5f * (zk11 + zk11);
    const float gCP0 = h00 * s.Cl0 + h01 * s.Cl1, gCP1 = h01 * s.Cl0 + h11 * s.Cl1;   // gCP = gS0 C (lane j: column j)
    const Vec4 gCP0v = spread(gCP0), gCP1v = spread(gCP1);
    // gSp += Z^T C + C^T gCP
    gSp = outer2(gCP0, s.Cl0, gCP1, s.Cl1, outer2(s.Cl0, z0, s.Cl1, z1, gSp));
    // gC = -K^T gIKC + Z Sig_p + gS0 (C Sig_p) + gCP Sig_p^T - gr mu_p^T    (lane j: column j)
    const float cp0 = dot(s.Spt, s.C0, 0.0f), cp1 = dot(s.Spt, s.C1, 0.0f);
    const float gC0 = -dot(gIKCt, k0v, 0.0f) + dot(s.Spt, spread(z0), 0.0f) + (h00 * cp0 + h01 * cp1) + dot(s.Sp, gCP0v, 0.0f) -
                      gr0 * s.mup;
    const float gC1 = -dot(gIKCt, k1v, 0.0f) + dot(s.Spt, spread(z1), 0.0f) + (h01 * cp0 + h11 * cp1) + dot(s.Sp, gCP1v, 0.0f) -
                      gr1 * s.mup;
    float *gCo = gstack_at(G.gC, b, t);
    gCo[i] = gC0, gCo[4 + i] = gC1;
    const float gmp = gmu + s.wmp - (s.Cl0 * gr0 + s.Cl1 * gr1);      // gmp = gmu + handed-off - C^T gr
    G.gY[q * 2 + (i & 1)] = (i & 1) ? gr1 : gr0;
    if constexpr (HAS_GQ) q4::store_rows(gstack_at(G.gQ, b, t), gSp, i);
    // gA[t] = smoother share + gSp^T (A Sig) + (gSp A) Sig^T + gmp mu^T ; carried adjoints for t-1
    const Mat gSpt = P(gSp, I4);
    const Mat ASt = P(s.A, s.Sigt);                                   // (A Sig)^T
    const Mat gAS = P(s.At, gSp), gASt = P(gSp, s.At);                // gSp A and its transpose
    Mat gA;
#pragma unroll
    for (int c = 0; c < 4; ++c) gA.c[c] = fmaf(gmp, s.mu.c[c], s.gAs.c[c]);
    gA = P(s.Sig, gAS, gA);                                           // + (gSp A) Sig^T
    gA = P(ASt, gSpt, gA);                                            // + gSp^T (A Sig)
    q4::store_rows(gstack_at(G.gA, b, t), gA, i);
    gSig = P(gASt, s.At), gSigt = P(s.At, gASt);                      // A^T (gSp A) and its transpose
    const Vec4 gmpv = spread(gmp);
    gmu = dot(s.At, gmpv, 0.0f);                                      // A^T gmp
    Mat gB;
#pragma unroll
    for (int c = 0; c < 4; ++c) gB.c[c] = gmp * s.u.c[c];
    q4::store_rows(gstack_at(G.gB, b, t), gB, i);
    G.gU[q * 4 + i] = dot(s.Bt, gmpv, 0.0f);                          // B^T gmp
  };
  int t = T - 1;
  for (; t >= 1; t -= 2) {
    step(t, s, nx);
    step(t - 1, nx, s);
  }
  if (t >= 0) step(t, s, nx);
  if (G.g_mu0) G.g_mu0[(int64_t)b * 4 + i] = gmu;
  if (G.g_Sigma0) q4::store_rows(G.g_Sigma0 + (int64_t)b * 16, gSig, i);
}


// ---- the kernels' bodies (kvae_lgssm_n16.hip wraps them in __global__ functions; tests/hostsim runs them on emulated wavefronts):
// sixteen sequences per wavefront, grid = ceil(B / 16); a ragged last wavefront recomputes (and re-stores, identically) the last
// sequence: no branch
// do_rts: 0 no smoother; 1 smoother; KV_M4_RTS_WITH_GAINS: smoother whose gains J_t are already in their slots (gains_wave)
#define KV_M4_RTS_WITH_GAINS 2
template <bool AUX>
__device__ __forceinline__ void smooth_fwd_wave(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S, int do_filter, int do_rts) {
  const int lane = threadIdx.x & 63, i = lane & 3;
  int b = blockIdx.x * 16 + (lane >> 2);
  b = b < P_.B ? b : P_.B - 1;
  if (do_filter && do_rts) {       // the filter sweep leaves the smoother gains behind (HOIST)
    filter_sweep<AUX, true>(P_, S, b, i, lane);
    __syncthreads();
    rts_sweep<AUX, true>(P_, S, b, i, lane);
  } else if (do_filter) {
    filter_sweep<AUX, false>(P_, S, b, i, lane);
  } else if (do_rts == KV_M4_RTS_WITH_GAINS) {
    rts_sweep<AUX, true>(P_, S, b, i, lane);
  } else if (do_rts) {
    rts_sweep<AUX, false>(P_, S, b, i, lane);
  }
}
// The smoother gains of ALL steps at once: J_t = Sig_f[t] A_{t+1}^T Sig_p[t+1]^{-1} depends on the filter's results only, so
// between a filter-only sweep and a gains-present smoother sweep it is B (T - 1) independent 4x4 problems - sixteen per wavefront,
// grid = ceil(B (T - 1) / 16) - instead of ~120 instructions inside every step of a T-deep dependent stream.  The launcher takes
// this three-launch form while the batch is far below the chip's wave slots (kv_m4_split: the sweeps are then bound by the
// length of their instruction stream); above that the single launch with the gain hoisted into the filter step moves fewer bytes.
constexpr int KV_M4_SPLIT_MAX_B = 2048;
inline int &kv_m4_split_override() {   // < 0: none (tests/hostsim sets it to compare the two forms in one process)
  static int v = -1;
  return v;
}
inline int kv_m4_split_max_b() {   // host side (launchers); KVAE_M4_SPLIT_MAX_B overrides (A/B runs; 0: always one launch)
  static const int v = getenv("KVAE_M4_SPLIT_MAX_B") ? atoi(getenv("KVAE_M4_SPLIT_MAX_B")) : KV_M4_SPLIT_MAX_B;
  return kv_m4_split_override() >= 0 ? kv_m4_split_override() : v;
}
inline bool kv_m4_split(const kvae_lgssm_problem &P_, int do_filter, int do_rts) {
  return do_filter && do_rts && P_.T >= 2 && P_.B <= kv_m4_split_max_b();
}
inline unsigned kv_m4_gain_grid(const kvae_lgssm_problem &P_) {
  return (unsigned)(((int64_t)P_.B * (P_.T - 1) + 15) / 16);
}
inline unsigned kv_m4_item_grid(const kvae_lgssm_problem &P_) { return (unsigned)(((int64_t)P_.B * P_.T + 15) / 16); }
template <bool AUX>
__device__ __forceinline__ void gains_wave(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S) {
  const int lane = threadIdx.x & 63, i = lane & 3;
  const int64_t items = (int64_t)P_.B * (P_.T - 1);
  int64_t it = (int64_t)blockIdx.x * 16 + (lane >> 2);
  it = it < items ? it : items - 1;      // a ragged last wavefront recomputes the last item
  const int b = (int)(it / (P_.T - 1)), t = (int)(it - (int64_t)b * (P_.T - 1));
  const int64_t q = (int64_t)b * P_.T + t;
  const Mat A = q4::load_rows(stack_at(P_.A, b, t + 1), i), Sf = q4::load_rows(S.Sigmas_filt + q * 16, i);
  const Mat Spt = load_cols(S.Sigmas_pred + (q + 1) * 16, i);
  const Mat W = P(Sf, A);                                    // A Sig_f   (Sig_f symmetric)
  bool bad;
  Mat X = solve_natural(Spt, W, i, bad);                     // Sig_p^T J^T = A Sig_f  (kalman_filter.py:229)
  if (__any(bad)) X = solve_pivoted(Spt, W, i, lane);
  q4::store_rows(gain_slot<AUX>(S, q), P(X, q4::eye(i)), i);
}
// ---- the filter's adjoint in two parts (same idea, same gate) -----------------------------------------------------------------------
// On the chain of filter_bwd_sweep: the adjoint of the filtered belief through the Joseph update and the gain (G -> X1 -> gIKC ->
// gK -> Z -> g Sig_p) and back through the prediction (g Sig_p -> gSp A -> A^T (gSp A)).  Off it: gC, gY, gA, gB, gU - a third
// of the instructions and nearly half of the products.  filter_bwd_chain leaves what those need in the hand-off record of the
// step it has just consumed ([ gmu | Gm | gmp | Z ]) and parks g Sig_p in the gB slot; filter_bwd_items reads them back for all
// (b, t) at once.  Same operations on the same operands as filter_bwd_sweep.
template <bool HAS_GQ>
__device__ __forceinline__ void filter_bwd_chain(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S,
                                                 const kvae_lgssm_input_grads &G, float *ws, int b, int i) {
  constexpr int WS_REC = q4::WS_REC;
  const int T = P_.T;
  const int64_t bT = (int64_t)b * T;
  const float R00 = P_.R[0], R01 = P_.R[1], R10 = P_.R[2], R11 = P_.R[3];
  const Mat I4 = q4::eye(i);
  struct In {
    Mat At, Sp, Spt, wSf, wSft, wSp;
    Vec4 C0, C1;
    float Cl0, Cl1, mk, mup, wmf, wmp;
    f2 y, ku;
    f4 Sv;
  } s, nx;
  q4::StepPtr ptr;
  ptr.init(P_, b, T - 1);
  int t_ld = T - 1;
  auto load = [&](In &o) {
    const int64_t q = bT + t_ld;
    o.At = load_cols(ptr.A, i);
    o.C0 = load_vec(ptr.C), o.C1 = load_vec(ptr.C + 4);
    o.Cl0 = ptr.C[i], o.Cl1 = ptr.C[4 + i];
    o.y = *reinterpret_cast<const f2 *>(ptr.Y);
    o.mk = *ptr.mk;
    o.Sp = q4::load_rows(S.Sigmas_pred + q * 16, i), o.Spt = load_cols(S.Sigmas_pred + q * 16, i);
    o.mup = S.mus_pred[q * 4 + i];
    const float *ax = S.aux + q * KV_AUX_N4;
    o.ku = *reinterpret_cast<const f2 *>(ax + 2 * i);
    o.Sv = *reinterpret_cast<const f4 *>(ax + 8);
    const float *w = ws + q * WS_REC;
    o.wmf = w[i];
    o.wSf = q4::load_rows(w + 4, i), o.wSft = load_cols(w + 4, i);
    o.wmp = w[4 + 16 + i];
    o.wSp = q4::load_rows(w + 4 + 16 + 4, i);
  };
  load(s);
  nx = s;
  KV_Q4_DRAIN();
  float gmu = 0.0f;
  Mat gSig = q4::zero(), gSigt = q4::zero();
  auto step = [&](int t, const In &s, In &nx) {
    if (t >= 1) t_ld = t - 1, ptr.step(-1);
    load(nx);
    KV_Q4_FENCE();
    const int64_t q = bT + t;
    const float mk = P_.mask ? s.mk : 1.0f;
    gmu += s.wmf;
    gSig = q4::add(gSig, s.wSf), gSigt = q4::add(gSigt, s.wSft);
    const Mat Gm = half_sum(gSig, gSigt);
    const float k0 = mk * s.ku[0], k1 = mk * s.ku[1];
    const Vec4 k0v = spread(k0), k1v = spread(k1);
    const Mat Mt = outer2(-k0, s.Cl0, -k1, s.Cl1, I4);                // (I - K C)^T = I - C0 k0^T - C1 k1^T
    const float gr0 = q4::qsum(k0 * gmu), gr1 = q4::qsum(k1 * gmu);   // gr = K^T gmu
    const float r0 = s.y[0] - q4::qsum(s.Cl0 * s.mup), r1 = s.y[1] - q4::qsum(s.Cl1 * s.mup);
    const Mat Sp2 = q4::add(s.Sp, s.Spt);                             // symmetric
    const Mat X1 = P(Mt, Gm), X1t = P(Gm, Mt);                        // G (I - K C) and its transpose
    const Mat gIKC = P(Sp2, X1);                                      // X1 (Sig_p^T + Sig_p)
    Mat gSp = P(X1t, Mt, s.wSp);                                      // (I - K C)^T X1 + handed-off
    const float GK0 = dot(Gm, k0v, 0.0f), GK1 = dot(Gm, k1v, 0.0f);
    const float gK0 = GK0 * (R00 + R00) + GK1 * (R01 + R10) - dot(gIKC, s.C0, 0.0f) + gmu * r0;
    const float gK1 = GK0 * (R10 + R01) + GK1 * (R11 + R11) - dot(gIKC, s.C1, 0.0f) + gmu * r1;
    const q4::Inv2 F = q4::factor2(s.Sv[0], s.Sv[1], s.Sv[3]);
    float z0, z1;
    q4::solve2(F, mk * gK0, mk * gK1, z0, z1);                        // Z = S^{-T} (mask gK^T): column i on lane i
    const float zk00 = q4::qsum(z0 * s.ku[0]), zk01 = q4::qsum(z0 * s.ku[1]), zk10 = q4::qsum(z1 * s.ku[0]),
                zk11 = q4::qsum(z1 * s.ku[1]);
    const float h00 = -0.5f * (zk00 + zk00), h01 = -0.5f * (zk01 + zk10), h11 = -0.5f * (zk11 + zk11);
    const float gCP0 = h00 * s.Cl0 + h01 * s.Cl1, gCP1 = h01 * s.Cl0 + h11 * s.Cl1;   // gCP = gS0 C (lane j: column j)
    gSp = outer2(gCP0, s.Cl0, gCP1, s.Cl1, outer2(s.Cl0, z0, s.Cl1, z1, gSp));   // gSp += Z^T C + C^T gCP
    const float gmp = gmu + s.wmp - (s.Cl0 * gr0 + s.Cl1 * gr1);      // gmp = gmu + handed-off - C^T gr
    // for filter_bwd_items: [ gmu | Gm | gmp | Z ] over the record this step has consumed, g Sig_p in the gB slot
    float *w = ws + q * WS_REC;
    w[i] = gmu;
    q4::store_rows(w + 4, Gm, i);
    w[4 + 16 + i] = gmp;
    w[4 + 16 + 4 + i] = z0, w[4 + 16 + 8 + i] = z1;
    q4::store_rows(gstack_at(G.gB, b, t), gSp, i);
    if constexpr (HAS_GQ) q4::store_rows(gstack_at(G.gQ, b, t), gSp, i);
    const Mat gAS = P(s.At, gSp), gASt = P(gSp, s.At);                // gSp A and its transpose
    gSig = P(gASt, s.At), gSigt = P(s.At, gASt);                      // A^T (gSp A) and its transpose
    gmu = dot(s.At, spread(gmp), 0.0f);                               // A^T gmp
    (void)gAS;
  };
  int t = T - 1;
  for (; t >= 1; t -= 2) {
    step(t, s, nx);
    step(t - 1, nx, s);
  }
  if (t >= 0) step(t, s, nx);
  if (G.g_mu0) G.g_mu0[(int64_t)b * 4 + i] = gmu;
  if (G.g_Sigma0) q4::store_rows(G.g_Sigma0 + (int64_t)b * 16, gSig, i);
}

__device__ __forceinline__ void filter_bwd_items(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S,
                                                 const kvae_lgssm_input_grads &G, const float *ws) {
  constexpr int WS_REC = q4::WS_REC;
  const int lane = threadIdx.x & 63, i = lane & 3;
  const int64_t items = (int64_t)P_.B * P_.T;
  int64_t q = (int64_t)blockIdx.x * 16 + (lane >> 2);
  q = q < items ? q : items - 1;         // a ragged last wavefront recomputes the last item
  const int b = (int)(q / P_.T), t = (int)(q - (int64_t)b * P_.T);
  const Mat I4 = q4::eye(i);
  q4::StepPtr ptr;
  ptr.init(P_, b, t);
  const Mat A = q4::load_rows(ptr.A, i), At = load_cols(ptr.A, i), Bt = load_cols(ptr.Bm, i);
  const Vec4 C0 = load_vec(ptr.C), C1 = load_vec(ptr.C + 4), u = load_vec(ptr.U);
  const float Cl0 = ptr.C[i], Cl1 = ptr.C[4 + i];
  const float mk = P_.mask ? *ptr.mk : 1.0f;
  const float *pS = t > 0 ? S.Sigmas_filt + (q - 1) * 16 : P_.Sigma0 + (int64_t)b * P_.Sigma0_sb;
  const float *pm = t > 0 ? S.mus_filt + (q - 1) * 4 : P_.mu0 + (int64_t)b * P_.mu0_sb;
  const Mat Sig = q4::load_rows(pS, i), Sigt = load_cols(pS, i);
  const Vec4 mu = load_vec(pm);
  const Mat Sp = q4::load_rows(S.Sigmas_pred + q * 16, i), Spt = load_cols(S.Sigmas_pred + q * 16, i);
  const float mup = S.mus_pred[q * 4 + i];
  const f2 ku = *reinterpret_cast<const f2 *>(S.aux + q * KV_AUX_N4 + 2 * i);
  const float *w = ws + q * WS_REC;
  const float gmu = w[i], gmp = w[4 + 16 + i], z0 = w[4 + 16 + 4 + i], z1 = w[4 + 16 + 8 + i];
  const Mat Gm = q4::load_rows(w + 4, i);
  float *gBo = gstack_at(G.gB, b, t), *gAo = gstack_at(G.gA, b, t);
  const Mat gSp = q4::load_rows(gBo, i), gAs = q4::load_rows(gAo, i);
  const float k0 = mk * ku[0], k1 = mk * ku[1];
  const Vec4 k0v = spread(k0), k1v = spread(k1);
  const Mat Mt = outer2(-k0, Cl0, -k1, Cl1, I4);
  const float gr0 = q4::qsum(k0 * gmu), gr1 = q4::qsum(k1 * gmu);     // gr = K^T gmu
  const Mat Sp2 = q4::add(Sp, Spt);
  const Mat X1 = P(Mt, Gm);
  const Mat gIKCt = P(X1, Sp2);                                       // (X1 (Sig_p^T + Sig_p))^T
  const float zk00 = q4::qsum(z0 * ku[0]), zk01 = q4::qsum(z0 * ku[1]), zk10 = q4::qsum(z1 * ku[0]), zk11 = q4::qsum(z1 * ku[1]);
  const float h00 = -0.5f * (zk00 + zk00), h01 = -0.5f * (zk01 + zk10), h11 = -0.5f * (zk11 + zk11);
  const float gCP0 = h00 * Cl0 + h01 * Cl1, gCP1 = h01 * Cl0 + h11 * Cl1;
  const Vec4 gCP0v = spread(gCP0), gCP1v = spread(gCP1);
  // gC = -K^T gIKC + Z Sig_p + gS0 (C Sig_p) + gCP Sig_p^T - gr mu_p^T    (lane j: column j)
  const float cp0 = dot(Spt, C0, 0.0f), cp1 = dot(Spt, C1, 0.0f);
  const float gC0 = -dot(gIKCt, k0v, 0.0f) + dot(Spt, spread(z0), 0.0f) + (h00 * cp0 + h01 * cp1) + dot(Sp, gCP0v, 0.0f) - gr0 * mup;
  const float gC1 = -dot(gIKCt, k1v, 0.0f) + dot(Spt, spread(z1), 0.0f) + (h01 * cp0 + h11 * cp1) + dot(Sp, gCP1v, 0.0f) - gr1 * mup;
  float *gCo = gstack_at(G.gC, b, t);
  gCo[i] = gC0, gCo[4 + i] = gC1;
  G.gY[q * 2 + (i & 1)] = (i & 1) ? gr1 : gr0;
  // gA[t] = smoother share + gSp^T (A Sig) + (gSp A) Sig^T + gmp mu^T
  const Mat gSpt = P(gSp, I4);
  const Mat ASt = P(A, Sigt);                                         // (A Sig)^T
  const Mat gAS = P(At, gSp);                                         // gSp A
  Mat gA;
#pragma unroll
  for (int c = 0; c < 4; ++c) gA.c[c] = fmaf(gmp, mu.c[c], gAs.c[c]);
  gA = P(Sig, gAS, gA);                                               // + (gSp A) Sig^T
  gA = P(ASt, gSpt, gA);                                              // + gSp^T (A Sig)
  q4::store_rows(gAo, gA, i);
  Mat gB;
#pragma unroll
  for (int c = 0; c < 4; ++c) gB.c[c] = gmp * u.c[c];
  q4::store_rows(gBo, gB, i);
  G.gU[q * 4 + i] = dot(Bt, spread(gmp), 0.0f);                       // B^T gmp
}

// part: KV_M4_BWD_ALL one launch; split form: KV_M4_BWD_CHAIN (the smoother adjoint's chain), rts_bwd_items, KV_M4_BWD_FCHAIN (the
// filter adjoint's chain), filter_bwd_items
#define KV_M4_BWD_ALL 0
#define KV_M4_BWD_CHAIN 1
#define KV_M4_BWD_FCHAIN 2
template <bool HAS_FP, bool HAS_GQ>
__device__ __forceinline__ void smooth_bwd_wave(const kvae_lgssm_problem &P_, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                                                const kvae_lgssm_input_grads &G, float *ws, int part) {
  const int lane = threadIdx.x & 63, i = lane & 3;
  int b = blockIdx.x * 16 + (lane >> 2);
  b = b < P_.B ? b : P_.B - 1;
  if (part == KV_M4_BWD_CHAIN) {
    rts_bwd_chain<HAS_FP>(P_, S, U, G, ws, b, i);
  } else if (part == KV_M4_BWD_FCHAIN) {
    filter_bwd_chain<HAS_GQ>(P_, S, G, ws, b, i);
  } else {
    rts_bwd_sweep<HAS_FP>(P_, S, U, G, ws, b, i, lane);
    __syncthreads();
    filter_bwd_sweep<HAS_GQ>(P_, S, G, ws, b, i, lane);
  }
}
inline bool kv_m4_split_bwd(const kvae_lgssm_problem &P_) { return P_.T >= 2 && P_.B <= kv_m4_split_max_b(); }   // host side

}  // namespace m4
}  // namespace kvae
#endif
