// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 3; DESIGN.md section 4 records the measurement): parity-green on the GPU (41 golden /
// oracle tests), and SLOWER than the quad layout it was meant to replace - filter 78 vs 69 us, RTS 85 vs 77 us at (B, T) =
// (32, 100); 43 / 45 vs 40 / 42 us at (256, 50).  A third of the vector instructions, but every product of two recursion-
// dependent matrices needs a cross-lane move through the LDS crossbar (ds_swizzle / ds_bpermute) IN the dependent chain, and
// five or six exposed ~130-cycle crossbar latencies per step cost more than the 160 DPP-folded FMAs they replace.
//
// lgssm_e4.h — filter + RTS smoother for (n, m, p) = (4, 4, 2) with ONE MATRIX ELEMENT PER LANE: a sequence owns a 16-lane
// DPP row, lane e = 4 i + j of the row holds element [i][j] of every 4x4 matrix in ONE register; four sequences per wavefront.
//
// Why (profiles/r03_q4_fwd_c4_sq.txt): the quad layout of lgssm_q4.h (row i of a matrix on lane i of a quad, sixteen sequences
// per wavefront) issues ~235 dependent vector instructions per filter step - 1029 cycles of issue plus 815 cycles of dependency
// stalls with the wavefront alone on its SIMD.  Below a few thousand sequences the chip has idle SIMDs for every wavefront, so
// the time of a step is the LENGTH of one wavefront's instruction stream, not its lane utilisation.  With an element per lane
//
//   C = A B        C_ij = sum_k A_ik B_kj :  four  v_fmac_f32_dpp  (A_ik: quad broadcast of the lane's own quad, folded into
//                                            the FMA) on four  ds_swizzle  results (B_kj: quad k of the row replicated over the
//                                            row - the LDS crossbar's bit-mask mode, no memory touched)      4 VALU, was 16
//   X^T            one ds_bpermute                                                                          0 VALU, was 16
//   sym(X)         one ds_bpermute + 2 VALU                                                                            was 24
//   A v, A^T v     multiply + two DPP adds (within the quad / across the quads of the row)                   3 VALU, was 4
//   outer products one FMA per term (column-vector x row-vector, both replicated along the other index)
//
// a filter step is ~70 vector instructions, and a quarter of the sequences per wavefront means four times the wavefronts to
// spread over the chip.  Same equations and op order as lgssm_q4.h / lgssm_fwd.h (reference kalman_filter.py:31-104, 204-237);
// same outputs and the same aux record (K unmasked | S | J), so the adjoint kernels of lgssm_q4.h take over unchanged.
// Operands are read with dword loads (no 16-byte alignment needed); per-step inputs are prefetched one step ahead.
#pragma once
#include "lgssm_q4.h"   // fmac_q / mul_q (DPP folded into the FMA), guard, frcp, factor2 / solve2, StepPtr, fences

#if !defined(KVAE_HOSTSIM) && !defined(KV_TPP)
namespace kvae {
namespace e4 {

using q4::fmac_q;
using q4::frcp;
using q4::guard;
using q4::mul_q;

// ---- cross-lane primitives inside one 16-lane row -----------------------------------------------------------------------------
// quad K of the row replicated over the row: lane (i, j) <- lane (K, j)
template <int K>
__device__ __forceinline__ float rowrep(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (K << 7) | 0x13));
}
// arbitrary gather inside the wavefront (byte address = 4 * source lane)
__device__ __forceinline__ float gather(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ float dppm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum / max over j (the lanes of a quad) and over i (the quads of the row: row_ror 4, 8); every lane gets the result
__device__ __forceinline__ float jsum(float x) { x += dppm<0xB1>(x); x += dppm<0x4E>(x); return x; }
__device__ __forceinline__ float isum(float x) { x += dppm<0x124>(x); x += dppm<0x128>(x); return x; }
__device__ __forceinline__ float imax(float x) { x = fmaxf(x, dppm<0x124>(x)); x = fmaxf(x, dppm<0x128>(x)); return x; }
__device__ __forceinline__ float imin(float x) { x = fminf(x, dppm<0x124>(x)); x = fminf(x, dppm<0x128>(x)); return x; }
template <int K> __device__ __forceinline__ float qbc(float v) { return dppm<K * 0x55>(v); }   // lane (i, j) <- lane (i, K)

struct Idx {
  int lane, i, j, tr;   // tr: the lane that holds the transposed element [j][i] of the same sequence
  __device__ __forceinline__ Idx() {
    lane = threadIdx.x & 63;
    i = (lane >> 2) & 3, j = lane & 3;
    tr = (lane & 48) | (j << 2) | i;
  }
};
// X^T; also: column-vector (v_i on every lane of quad i) <-> row-vector (v_j on lane j of every quad)
__device__ __forceinline__ float tpose(float X, const Idx &x) { return gather(X, x.tr); }

// C = A B (+ C0): the DPP source is A (callers guard() it if their own VALU code has just written it)
__device__ __forceinline__ float mul_nn(float A, float B) {
  const float b0 = rowrep<0>(B), b1 = rowrep<1>(B), b2 = rowrep<2>(B), b3 = rowrep<3>(B);
  float c = mul_q<0>(A, b0);
  fmac_q<1>(c, A, b1);
  fmac_q<2>(c, A, b2);
  fmac_q<3>(c, A, b3);
  return c;
}
__device__ __forceinline__ float mul_nn_acc(float A, float B, float C0) {
  const float b0 = rowrep<0>(B), b1 = rowrep<1>(B), b2 = rowrep<2>(B), b3 = rowrep<3>(B);
  float c = C0;
  fmac_q<0>(c, A, b0);
  fmac_q<1>(c, A, b1);
  fmac_q<2>(c, A, b2);
  fmac_q<3>(c, A, b3);
  return c;
}
__device__ __forceinline__ float symm(float X, const Idx &x) { return 0.5f * (X + tpose(X, x)); }

// ---- 4x4 solve  M X = W  (M, W one element per lane): Gauss-Jordan with partial pivoting, the pivot sequence and multipliers of
// torch.linalg.solve / getrf (first maximum of |column|); the row exchange is a gather taken only when some sequence needs it ----
template <int K>
__device__ __forceinline__ void gj_step(float &m, float &w, float &rinv_i, const Idx &x) {
  const float colK = qbc<K>(m);                                  // m[i][K] on every lane of quad i
  const float cand = x.i >= K ? fabsf(colK) : -1.0f;
  const float mx = imax(cand);
  const float p = imin((cand == mx) ? (float)x.i : 9.0f);       // first row holding the maximum (all-NaN column: 9 -> keep K)
  const int pi = p < 4.0f ? (int)p : K;
  if (__any(pi != K)) {                                          // some sequence of the wavefront exchanges rows K and p
    const int src_i = x.i == K ? pi : (x.i == pi ? K : x.i);
    const int src = (x.lane & 48) | (src_i << 2) | x.j;
    m = gather(m, src);
    w = gather(w, src);
  }
  const float rowK = rowrep<K>(m), wK = rowrep<K>(w);            // m[K][j], w[K][j]
  const float piv = qbc<K>(rowK);                                // m[K][K]
  const float rinv = frcp(piv);
  const float mik = qbc<K>(m);                                   // m[i][K] after the exchange
  const float f = x.i == K ? 0.0f : -(mik * rinv);
  rinv_i = x.i == K ? rinv : rinv_i;
  m = fmaf(f, rowK, m);
  w = fmaf(f, wK, w);
  if constexpr (K + 1 < 4) gj_step<K + 1>(m, w, rinv_i, x);
}
__device__ __forceinline__ float solve(float m, float w, const Idx &x) {
  float rinv_i = 0.0f;
  gj_step<0>(m, w, rinv_i, x);
  return w * rinv_i;
}

// ---- per-step operands ------------------------------------------------------------------------------------------------------------
struct StepIn {
  float A, At, Bm, Q, C0r, C1r, C0c, C1c, u, y0, y1, mk;   // C*r: C[.][j] (row-vector), C*c: C[.][i] (column-vector), u: u[j]
};
__device__ __forceinline__ void load_step(const q4::StepPtr &p, const Idx &x, StepIn &s) {
  const int e = (x.i << 2) | x.j, et = (x.j << 2) | x.i;
  s.A = p.A[e], s.At = p.A[et], s.Bm = p.Bm[e], s.Q = p.Q[e];
  s.C0r = p.C[x.j], s.C1r = p.C[4 + x.j], s.C0c = p.C[x.i], s.C1c = p.C[4 + x.i];
  s.u = p.U[x.j];
  s.y0 = p.Y[0], s.y1 = p.Y[1];
  s.mk = *p.mk;   // raw; the NULL-mask select happens at the point of use
}

template <bool AUX>
__device__ __forceinline__ void filter_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, const Idx &x) {
  const int T = P.T, e = (x.i << 2) | x.j;
  const int64_t bT = (int64_t)b * T;
  float Sig = (P.Sigma0 + (int64_t)b * P.Sigma0_sb)[e];
  float mu_r = P.mu0[(int64_t)b * P.mu0_sb + x.j];            // mu as a row-vector
  const float R00 = P.R[0], R01 = P.R[1], R10 = P.R[2], R11 = P.R[3];
  const float I4 = x.i == x.j ? 1.0f : 0.0f;
  q4::StepPtr ptr;
  ptr.init(P, b, 0);
  StepIn s, nx;
  load_step(ptr, x, s);
  nx = s;
  KV_Q4_DRAIN();
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) ptr.step(1);
    load_step(ptr, x, nx);                                   // unconditional prefetch, pinned above this step's stores
    KV_Q4_FENCE();
    const int64_t q = bT + t;
    // predict (kalman_filter.py:65-67)
    const float mup = jsum(fmaf(s.Bm, s.u, s.A * mu_r));     // column-vector: (A mu + B u)_i
    float Ag = s.A;                                          // (a register copy of the prefetched value may sit right in front)
    guard(Ag);
    const float AS = mul_nn(Ag, Sig);
    float ASg = AS;
    guard(ASg);
    const float Sigp = mul_nn_acc(ASg, s.At, s.Q);           // (A Sig) A^T + Q
    S.Sigmas_pred[q * 16 + e] = Sigp;
    S.mus_pred[q * 4 + x.i] = mup;
    // innovation (:73-90): PCT = Sigp C^T (column-vectors), S = sym(C PCT + R), r = y - C mup
    const float pct0 = jsum(Sigp * s.C0r), pct1 = jsum(Sigp * s.C1r);
    const float a00 = isum(s.C0c * pct0) + R00, a01 = isum(s.C0c * pct1) + R01;
    const float a10 = isum(s.C1c * pct0) + R10, a11 = isum(s.C1c * pct1) + R11;
    const float s00 = 0.5f * (a00 + a00), s01 = 0.5f * (a01 + a10), s11 = 0.5f * (a11 + a11);
    const float r0 = s.y0 - isum(s.C0c * mup), r1 = s.y1 - isum(s.C1c * mup);
    const q4::Inv2 F = q4::factor2(s00, s01, s11);
    float ku0, ku1;
    q4::solve2(F, pct0, pct1, ku0, ku1);                     // unmasked gain K[i][:] (column-vectors)
    if constexpr (AUX) {
      float *ax = S.aux + q * KV_AUX_N4;
      if (x.j < 2) ax[2 * x.i + x.j] = x.j == 0 ? ku0 : ku1;
      ax[8 + (e & 3)] = (e & 3) == 0 ? s00 : ((e & 3) == 3 ? s11 : s01);
    }
    const float mk = P.mask ? s.mk : 1.0f;
    const float k0 = mk * ku0, k1 = mk * ku1;                // :92
    const float muf = mup + k0 * r0 + k1 * r1;               // :96, column-vector
    S.mus_filt[q * 4 + x.i] = muf;
    // Joseph update (:97-101): M = I - K C ; (M Sigp) M^T + (K R) K^T
    const float k0r = tpose(k0, x), k1r = tpose(k1, x);      // the gain as row-vectors
    const float M = fmaf(-k1, s.C1r, fmaf(-k0, s.C0r, I4));
    const float Mt = fmaf(-s.C1c, k1r, fmaf(-s.C0c, k0r, I4));
    const float kr0 = k0 * R00 + k1 * R10, kr1 = k0 * R01 + k1 * R11;
    const float KRK = fmaf(kr1, k1r, kr0 * k0r);
    float Mg = M;
    guard(Mg);
    const float T1 = mul_nn(Mg, Sigp);
    float T1g = T1;
    guard(T1g);
    const float F0 = mul_nn_acc(T1g, Mt, KRK);
    Sig = symm(F0, x);
    S.Sigmas_filt[q * 16 + e] = Sig;
    mu_r = tpose(muf, x);
    s = nx;
  }
}

template <bool AUX>
__device__ __forceinline__ void rts_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, const Idx &x) {
  const int T = P.T, e = (x.i << 2) | x.j, et = (x.j << 2) | x.i;
  const int64_t bT = (int64_t)b * T;
  float SigS = S.Sigmas_filt[(bT + T - 1) * 16 + e];
  float mus = S.mus_filt[(bT + T - 1) * 4 + x.i];            // column-vector
  S.Sigmas_smooth[(bT + T - 1) * 16 + e] = SigS;
  S.mus_smooth[(bT + T - 1) * 4 + x.i] = mus;
  struct In { float Sf, Sp, Spt, A, muf, mup; } s, nx;
  int64_t q = bT + (T >= 2 ? T - 2 : 0);
  const float *pA = stack_at(P.A, b, T >= 2 ? T - 1 : 0);
  const int64_t sA = P.A.st;
  auto load = [&](In &o) {
    o.Sf = S.Sigmas_filt[q * 16 + e];
    o.Sp = S.Sigmas_pred[(q + 1) * 16 + e];
    o.Spt = S.Sigmas_pred[(q + 1) * 16 + et];
    o.A = pA[e];
    o.muf = S.mus_filt[q * 4 + x.i];
    o.mup = S.mus_pred[(q + 1) * 4 + x.i];
  };
  if (T >= 2) load(s);
  nx = s;
  KV_Q4_DRAIN();
  for (int t = T - 2; t >= 0; --t) {
    const int64_t qt = bT + t;
    if (t >= 1) q -= 1, pA -= sA;
    load(nx);
    KV_Q4_FENCE();
    // J = Sig_f A^T Sigp^{-1}  <=>  Sigp^T J^T = A Sig_f  (kalman_filter.py:229)
    float Ag = s.A;
    guard(Ag);
    const float W = mul_nn(Ag, s.Sf);
    const float Xt = solve(s.Spt, W, x);                     // J^T
    const float J = tpose(Xt, x);
    if constexpr (AUX) S.aux[qt * KV_AUX_N4 + 12 + e] = J;
    const float D = SigS - s.Sp;
    const float dmu_r = tpose(mus - s.mup, x);               // row-vector
    float Jg = J;
    guard(Jg);
    const float E = mul_nn(Jg, D);
    mus = s.muf + jsum(J * dmu_r);                           // :232
    float Eg = E;
    guard(Eg);
    const float Fm = mul_nn_acc(Eg, Xt, s.Sf);               // Sig_f + (J D) J^T   (:234)
    SigS = symm(Fm, x);
    S.Sigmas_smooth[qt * 16 + e] = SigS;
    S.mus_smooth[qt * 4 + x.i] = mus;
    s = nx;
  }
}

}  // namespace e4
}  // namespace kvae
#endif
