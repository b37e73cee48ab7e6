// lgssm_q4.h — filter / RTS smoother / their adjoints for (n, m, p) = (4, 4, 2) with SIXTEEN sequences per wavefront:
// a sequence owns one quad of lanes, lane i of the quad owns ROW i of every 4x4 matrix (4 VGPRs per matrix) and element i
// of every vector.  This is the rows-on-lanes layout of lgssm_n16.h shrunk to quads; the cross-lane traffic of every product,
// solve and reduction is a DPP quad_perm folded into the FMA (v_fmac_f32_dpp / v_mul_f32_dpp, spelled as inline asm because
// hipcc does not fold a DPP mov into an fma), so there is no LDS, no barrier and no shuffle anywhere:
//
//   C = A B    : C_i[c] = sum_k A_i[k] * (lane k's B[c])          16 instructions for 16 sequences
//   C = A B^T  : C_i[c] = sum_k A_i[k] * (lane c's B[k])          16
//   y = A v    : y_i    = sum_k A_i[k] * (lane k's v)              4
//   X^T        : two butterfly stages (quad_perm [2,3,0,1], [1,0,3,2] + selects)   16
//   4x4 solve  : Gauss-Jordan with partial pivoting, rows on lanes, pivot row by quad broadcast (same pivot sequence and
//                multipliers as torch.linalg.solve / getrf; the row exchange is a ds_bpermute, taken only when some quad needs it)
//
// The one-wavefront-per-sequence kernels of lgssm_n4.h keep at most 16 of 64 lanes busy and round-trip every phase through LDS;
// they remain for run-time m and as the reference the tests compare this file with.  Same equations, cited in lgssm_fwd.h /
// lgssm_bwd.h; same aux record (K unmasked | S | J) as lgssm_n4.h.
#pragma once
#include "lgssm_n4.h"   // stack_at, mask_addr, KV_AUX_N4

#if !defined(KVAE_HOSTSIM) && !defined(KV_TPP)
namespace kvae {
namespace q4 {

using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
struct Mat { float c[4]; };   // lane i: row i

// acc += (quad lane K of src) * f   /   (quad lane K of src) * f
#define KV_Q4_DPP(K)                                                                                                        \
  template <> __device__ __forceinline__ void fmac_q<K>(float &acc, float src, float f) {                                  \
    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf"            \
                 : "+v"(acc) : "v"(src), "v"(f));                                                                          \
  }                                                                                                                         \
  template <> __device__ __forceinline__ float mul_q<K>(float src, float f) {                                              \
    float r;                                                                                                                \
    asm volatile("v_mul_f32_dpp %0, %1, %2 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf"             \
                 : "=v"(r) : "v"(src), "v"(f));                                                                            \
    return r;                                                                                                               \
  }
template <int K> __device__ __forceinline__ void fmac_q(float &acc, float src, float f);
template <int K> __device__ __forceinline__ float mul_q(float src, float f);
KV_Q4_DPP(0) KV_Q4_DPP(1) KV_Q4_DPP(2) KV_Q4_DPP(3)
#undef KV_Q4_DPP

// Two wait states between a VALU write and a DPP read of the same register: the compiler's hazard recogniser cannot see
// into the asm above, so every value that the compiler's own VALU code produced and that is about to be read THROUGH DPP
// passes through one of these first.  The "+v" operands tie the s_nop into the data flow: the producer cannot be scheduled
// after it, the consumer not before it.  (Values produced by the asm itself are ordered by `volatile`.)
__device__ __forceinline__ void guard(float &a) { asm volatile("s_nop 1" : "+v"(a)); }
__device__ __forceinline__ void guard(float &a, float &b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }

template <int CTRL>
__device__ __forceinline__ float dppm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int K> __device__ __forceinline__ float qb(float v) { return dppm<K * 0x55>(v); }      // quad broadcast of lane K
__device__ __forceinline__ float qx1(float v) { return dppm<0xB1>(v); }                          // lane ^ 1
__device__ __forceinline__ float qx2(float v) { return dppm<0x4E>(v); }                          // lane ^ 2
__device__ __forceinline__ float qsum(float x) { x += qx1(x); x += qx2(x); return x; }           // over the quad, all lanes
__device__ __forceinline__ float qmax(float x) { x = fmaxf(x, qx1(x)); x = fmaxf(x, qx2(x)); return x; }

__device__ __forceinline__ void guard(Mat &m) { asm volatile("s_nop 1" : "+v"(m.c[0]), "+v"(m.c[1]), "+v"(m.c[2]), "+v"(m.c[3])); }

// ---- products: the DPP source is always the SECOND matrix / the vector; callers guard() it if their own VALU code made it --
// (the four accumulators of a product are advanced round-robin: consecutive instructions are independent, so the wavefront -
//  usually alone on its SIMD - issues them back to back instead of waiting out each FMA's latency)
__device__ __forceinline__ Mat mul_nn(const Mat &A, const Mat &B) {     // A B
  Mat C;
  C.c[0] = mul_q<0>(B.c[0], A.c[0]), C.c[1] = mul_q<0>(B.c[1], A.c[0]), C.c[2] = mul_q<0>(B.c[2], A.c[0]), C.c[3] = mul_q<0>(B.c[3], A.c[0]);
  fmac_q<1>(C.c[0], B.c[0], A.c[1]), fmac_q<1>(C.c[1], B.c[1], A.c[1]), fmac_q<1>(C.c[2], B.c[2], A.c[1]), fmac_q<1>(C.c[3], B.c[3], A.c[1]);
  fmac_q<2>(C.c[0], B.c[0], A.c[2]), fmac_q<2>(C.c[1], B.c[1], A.c[2]), fmac_q<2>(C.c[2], B.c[2], A.c[2]), fmac_q<2>(C.c[3], B.c[3], A.c[2]);
  fmac_q<3>(C.c[0], B.c[0], A.c[3]), fmac_q<3>(C.c[1], B.c[1], A.c[3]), fmac_q<3>(C.c[2], B.c[2], A.c[3]), fmac_q<3>(C.c[3], B.c[3], A.c[3]);
  return C;
}
__device__ __forceinline__ Mat mul_nn_acc(const Mat &A, const Mat &B, const Mat &C0) {     // A B + C0
  Mat C = C0;
  fmac_q<0>(C.c[0], B.c[0], A.c[0]), fmac_q<0>(C.c[1], B.c[1], A.c[0]), fmac_q<0>(C.c[2], B.c[2], A.c[0]), fmac_q<0>(C.c[3], B.c[3], A.c[0]);
  fmac_q<1>(C.c[0], B.c[0], A.c[1]), fmac_q<1>(C.c[1], B.c[1], A.c[1]), fmac_q<1>(C.c[2], B.c[2], A.c[1]), fmac_q<1>(C.c[3], B.c[3], A.c[1]);
  fmac_q<2>(C.c[0], B.c[0], A.c[2]), fmac_q<2>(C.c[1], B.c[1], A.c[2]), fmac_q<2>(C.c[2], B.c[2], A.c[2]), fmac_q<2>(C.c[3], B.c[3], A.c[2]);
  fmac_q<3>(C.c[0], B.c[0], A.c[3]), fmac_q<3>(C.c[1], B.c[1], A.c[3]), fmac_q<3>(C.c[2], B.c[2], A.c[3]), fmac_q<3>(C.c[3], B.c[3], A.c[3]);
  return C;
}
__device__ __forceinline__ Mat mul_nt(const Mat &A, const Mat &B, const Mat &C0) {   // A B^T + C0 : C_i[c] += A_i[k] (lane c's B[k])
  Mat C = C0;
  fmac_q<0>(C.c[0], B.c[0], A.c[0]), fmac_q<1>(C.c[1], B.c[0], A.c[0]), fmac_q<2>(C.c[2], B.c[0], A.c[0]), fmac_q<3>(C.c[3], B.c[0], A.c[0]);
  fmac_q<0>(C.c[0], B.c[1], A.c[1]), fmac_q<1>(C.c[1], B.c[1], A.c[1]), fmac_q<2>(C.c[2], B.c[1], A.c[1]), fmac_q<3>(C.c[3], B.c[1], A.c[1]);
  fmac_q<0>(C.c[0], B.c[2], A.c[2]), fmac_q<1>(C.c[1], B.c[2], A.c[2]), fmac_q<2>(C.c[2], B.c[2], A.c[2]), fmac_q<3>(C.c[3], B.c[2], A.c[2]);
  fmac_q<0>(C.c[0], B.c[3], A.c[3]), fmac_q<1>(C.c[1], B.c[3], A.c[3]), fmac_q<2>(C.c[2], B.c[3], A.c[3]), fmac_q<3>(C.c[3], B.c[3], A.c[3]);
  return C;
}
__device__ __forceinline__ float matvec(const Mat &A, float v, float acc) {          // acc + (A v)_i
  fmac_q<0>(acc, v, A.c[0]);
  fmac_q<1>(acc, v, A.c[1]);
  fmac_q<2>(acc, v, A.c[2]);
  fmac_q<3>(acc, v, A.c[3]);
  return acc;
}
// rank-one / rank-two pieces: acc[j] += a_i * (lane j's b)
__device__ __forceinline__ void outer_acc(Mat &M, float a, float b) {
  fmac_q<0>(M.c[0], b, a);
  fmac_q<1>(M.c[1], b, a);
  fmac_q<2>(M.c[2], b, a);
  fmac_q<3>(M.c[3], b, a);
}
__device__ __forceinline__ Mat transpose(const Mat &X, int i) {
  // The DPP moves are evaluated by ALL lanes before any select: inside an arm of `cond ? a : dpp(b)` they would run under a
  // partial EXEC mask, and a DPP read of an inactive lane returns 0.
  const bool lo = i < 2, ev = (i & 1) == 0;
  const float x0 = qx2(X.c[0]), x1 = qx2(X.c[1]), x2 = qx2(X.c[2]), x3 = qx2(X.c[3]);
  Mat Y, Z;
  Y.c[0] = lo ? X.c[0] : x2;
  Y.c[1] = lo ? X.c[1] : x3;
  Y.c[2] = lo ? x0 : X.c[2];
  Y.c[3] = lo ? x1 : X.c[3];
  const float y0 = qx1(Y.c[0]), y1 = qx1(Y.c[1]), y2 = qx1(Y.c[2]), y3 = qx1(Y.c[3]);
  Z.c[0] = ev ? Y.c[0] : y1;
  Z.c[1] = ev ? y0 : Y.c[1];
  Z.c[2] = ev ? Y.c[2] : y3;
  Z.c[3] = ev ? y2 : Y.c[3];
  return Z;
}
__device__ __forceinline__ Mat symmetrise(const Mat &X, int i) {
  const Mat T = transpose(X, i);
  Mat S;
#pragma unroll
  for (int c = 0; c < 4; ++c) S.c[c] = 0.5f * (X.c[c] + T.c[c]);
  return S;
}
__device__ __forceinline__ Mat sub(const Mat &A, const Mat &B) {
  Mat C;
#pragma unroll
  for (int c = 0; c < 4; ++c) C.c[c] = A.c[c] - B.c[c];
  return C;
}
__device__ __forceinline__ Mat add(const Mat &A, const Mat &B) {
  Mat C;
#pragma unroll
  for (int c = 0; c < 4; ++c) C.c[c] = A.c[c] + B.c[c];
  return C;
}
__device__ __forceinline__ Mat zero() { return Mat{{0.f, 0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Mat eye(int i) { return Mat{{i == 0 ? 1.f : 0.f, i == 1 ? 1.f : 0.f, i == 2 ? 1.f : 0.f, i == 3 ? 1.f : 0.f}}; }

__device__ __forceinline__ Mat load_rows(const float *X, int i) {
  const f4 v = *reinterpret_cast<const f4 *>(X + 4 * i);
  return Mat{{v[0], v[1], v[2], v[3]}};
}
__device__ __forceinline__ void store_rows(float *X, const Mat &M, int i) {
  *reinterpret_cast<f4 *>(X + 4 * i) = f4{M.c[0], M.c[1], M.c[2], M.c[3]};
}
__device__ __forceinline__ float frcp(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// ---- 4x4 solve  Mt X = W  (rows of Mt and of W on the lanes): Gauss-Jordan, partial pivoting per quad -------------------
template <int K>
__device__ __forceinline__ void gj_step(Mat &m, Mat &x, float &my_rinv, int i, int lane) {
  const float cand = i >= K ? fabsf(m.c[K]) : -1.0f;
  const float mx = qmax(cand);
  const unsigned long long hit = __ballot(cand == mx);
  const unsigned nib = (unsigned)(hit >> (lane & 60)) & 0xfu;          // this quad's candidates
  const int p = nib ? __builtin_ctz(nib) : K;                          // first maximum wins; all-NaN column: keep the diagonal
  if (__any(p != K)) {                                                  // some quad exchanges rows K and p
    const int src = (lane & 60) | (i == K ? p : (i == p ? K : i));
#pragma unroll
    for (int c = K; c < 4; ++c) m.c[c] = __shfl(m.c[c], src, 64);
#pragma unroll
    for (int c = 0; c < 4; ++c) x.c[c] = __shfl(x.c[c], src, 64);
  }
  const float piv = qb<K>(m.c[K]);
  const float rinv = frcp(piv);
  const float f = i == K ? 0.0f : -(m.c[K] * rinv);
  if (i == K) my_rinv = rinv;
  guard(m);     // rows may come from the compiler's own code (transpose selects, the exchange above)
  guard(x);
#pragma unroll
  for (int c = K + 1; c < 4; ++c) fmac_q<K>(m.c[c], m.c[c], f);
#pragma unroll
  for (int c = 0; c < 4; ++c) fmac_q<K>(x.c[c], x.c[c], f);
  if constexpr (K + 1 < 4) gj_step<K + 1>(m, x, my_rinv, i, lane);
}
__device__ __forceinline__ Mat solve(Mat m, Mat x, int i, int lane) {
  float my_rinv = 0.0f;
  gj_step<0>(m, x, my_rinv, i, lane);
#pragma unroll
  for (int c = 0; c < 4; ++c) x.c[c] *= my_rinv;
  return x;
}

// x = S^{-1} b for the symmetric 2x2 innovation covariance, partial pivoting (as lu_solve / getrf)
struct Inv2 { float a01, r00, l, ru11; bool sw; };
__device__ __forceinline__ Inv2 factor2(float s00, float s01, float s11) {
  Inv2 o;
  o.sw = fabsf(s01) > fabsf(s00);
  const float a00 = o.sw ? s01 : s00, a10 = o.sw ? s00 : s01, a11 = o.sw ? s01 : s11;
  o.a01 = o.sw ? s11 : s01;
  o.r00 = frcp(a00);
  o.l = a10 * o.r00;
  o.ru11 = frcp(fmaf(-o.l, o.a01, a11));
  return o;
}
__device__ __forceinline__ void solve2(const Inv2 &F, float b0, float b1, float &x0, float &x1) {
  const float c0 = F.sw ? b1 : b0, c1 = F.sw ? b0 : b1;
  x1 = fmaf(-F.l, c0, c1) * F.ru11;
  x0 = fmaf(-F.a01, x1, c0) * F.r00;
}

// ---- per-lane running pointers of one sequence -------------------------------------------------------------------------
struct StepPtr {
  const float *A, *Bm, *Q, *C, *U, *Y, *mk;
  int64_t sA, sB, sQ, sC, smk;
  __device__ __forceinline__ void init(const kvae_lgssm_problem &P, int b, int t) {
    A = stack_at(P.A, b, t), Bm = stack_at(P.Bm, b, t), Q = stack_at(P.Q, b, t), C = stack_at(P.C, b, t);
    sA = P.A.st, sB = P.Bm.st, sQ = P.Q.st, sC = P.C.st;
    U = P.U + ((int64_t)b * P.T + t) * 4, Y = P.Y + ((int64_t)b * P.T + t) * 2;
    mk = P.mask ? P.mask + (int64_t)b * P.T + t : P.R;
    smk = P.mask ? 1 : 0;
  }
  __device__ __forceinline__ void step(int d) { A += d * sA, Bm += d * sB, Q += d * sQ, C += d * sC, U += d * 4, Y += d * 2, mk += d * smk; }
};
struct StepIn {
  Mat A, Bm, Q;
  float Cl0, Cl1, u, mk;
  f2 y;
};
__device__ __forceinline__ void load_step(const StepPtr &p, int i, StepIn &s) {
  s.A = load_rows(p.A, i), s.Bm = load_rows(p.Bm, i), s.Q = load_rows(p.Q, i);
  s.Cl0 = p.C[i], s.Cl1 = p.C[4 + i];
  s.u = p.U[i];
  s.y = *reinterpret_cast<const f2 *>(p.Y);
  s.mk = *p.mk;   // raw; the NULL-mask select happens at the point of use (see lgssm_n16.h)
}
#define KV_Q4_FENCE() asm volatile("" ::: "memory")
#define KV_Q4_DRAIN()                      \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_waitcnt(0x0F70);    \
    asm volatile("" ::: "memory");         \
  } while (0)

// ---- forward sweeps ------------------------------------------------------------------------------------------------------
template <bool AUX>
__device__ __forceinline__ void filter_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, int i) {
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  Mat Sig = load_rows(P.Sigma0 + (int64_t)b * P.Sigma0_sb, i);
  float mu = P.mu0[(int64_t)b * P.mu0_sb + i];
  const float R00 = P.R[0], R01 = P.R[1], R10 = P.R[2], R11 = P.R[3];
  const Mat I4 = eye(i);
  StepPtr ptr;
  ptr.init(P, b, 0);
  StepIn s, nx;
  load_step(ptr, i, s);
  nx = s;
  KV_Q4_DRAIN();
  for (int t = 0; t < T; ++t) {
    if (t + 1 < T) ptr.step(1);
    load_step(ptr, i, nx);                                   // unconditional prefetch, pinned above this step's stores
    KV_Q4_FENCE();
    const int64_t q = bT + t;
    guard(mu);
    guard(Sig);
    // predict (kalman_filter.py:65-67)
    const float mup = matvec(s.Bm, s.u, matvec(s.A, mu, 0.0f));
    const Mat AS = mul_nn(s.A, Sig);
    const Mat Sigp = mul_nt(AS, s.A, s.Q);                   // (A Sig) A^T + Q
    store_rows(S.Sigmas_pred + q * 16, Sigp, i);
    S.mus_pred[q * 4 + i] = mup;
    // innovation (:73-90): PCT = Sigp C^T (lane i: PCT[i][c]), S = sym(C PCT + R), r = y - C mup
    const float pct0 = matvec(Sigp, s.Cl0, 0.0f), pct1 = matvec(Sigp, s.Cl1, 0.0f);
    const float a00 = qsum(s.Cl0 * pct0) + R00, a01 = qsum(s.Cl0 * pct1) + R01;
    const float a10 = qsum(s.Cl1 * pct0) + R10, a11 = qsum(s.Cl1 * pct1) + R11;
    const float s00 = 0.5f * (a00 + a00), s01 = 0.5f * (a01 + a10), s11 = 0.5f * (a11 + a11);
    const float r0 = s.y[0] - qsum(s.Cl0 * mup), r1 = s.y[1] - qsum(s.Cl1 * mup);
    const Inv2 F = factor2(s00, s01, s11);
    float ku0, ku1;
    solve2(F, pct0, pct1, ku0, ku1);                         // unmasked gain K[i][:]
    if constexpr (AUX) {
      float *ax = S.aux + q * KV_AUX_N4;
      *reinterpret_cast<f2 *>(ax + 2 * i) = f2{ku0, ku1};
      *reinterpret_cast<f4 *>(ax + 8) = f4{s00, s01, s01, s11};
    }
    const float mk = P.mask ? s.mk : 1.0f;
    const float k0 = mk * ku0, k1 = mk * ku1;                // :92
    const float muf = mup + k0 * r0 + k1 * r1;               // :96
    S.mus_filt[q * 4 + i] = muf;
    // Joseph update (:97-101): M = I - K C ; (M Sigp) M^T + (K R) K^T
    Mat M = I4;
    const float nk0 = -k0, nk1 = -k1;
    outer_acc(M, nk0, s.Cl0);
    outer_acc(M, nk1, s.Cl1);
    const float kr0 = k0 * R00 + k1 * R10, kr1 = k0 * R01 + k1 * R11;
    Mat KRK = zero();
    float k0g = k0, k1g = k1;
    guard(k0g, k1g);
    outer_acc(KRK, kr0, k0g);
    outer_acc(KRK, kr1, k1g);
    const Mat T1 = mul_nn(M, Sigp);
    const Mat F0 = mul_nt(T1, M, KRK);
    Sig = symmetrise(F0, i);
    store_rows(S.Sigmas_filt + q * 16, Sig, i);
    mu = muf;
    s = nx;
  }
}

template <bool AUX>
__device__ __forceinline__ void rts_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, int i, int lane) {
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  Mat SigS = load_rows(S.Sigmas_filt + (bT + T - 1) * 16, i);
  float mus = S.mus_filt[(bT + T - 1) * 4 + i];
  store_rows(S.Sigmas_smooth + (bT + T - 1) * 16, SigS, i);
  S.mus_smooth[(bT + T - 1) * 4 + i] = mus;
  struct In { Mat Sf, Sp, A; float muf, mup; } s, nx;
  int64_t q = bT + (T >= 2 ? T - 2 : 0);
  const float *pA = stack_at(P.A, b, T >= 2 ? T - 1 : 0);
  const int64_t sA = P.A.st;
  auto load = [&](In &o) {
    o.Sf = load_rows(S.Sigmas_filt + q * 16, i);
    o.Sp = load_rows(S.Sigmas_pred + (q + 1) * 16, i);
    o.A = load_rows(pA, i);
    o.muf = S.mus_filt[q * 4 + i];
    o.mup = S.mus_pred[(q + 1) * 4 + i];
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
    const Mat W = mul_nn(s.A, s.Sf);
    const Mat Spt = transpose(s.Sp, i);
    const Mat X = solve(Spt, W, i, lane);                    // J^T, rows on lanes
    Mat J = transpose(X, i);
    if constexpr (AUX) store_rows(S.aux + qt * KV_AUX_N4 + 12, J, i);
    Mat D = sub(SigS, s.Sp);
    float dmu = mus - s.mup;
    guard(D);
    guard(dmu);
    guard(J);
    const Mat E = mul_nn(J, D);
    mus = matvec(J, dmu, s.muf);                             // :232
    const Mat Fm = mul_nt(E, J, s.Sf);                       // Sig_f + (J D) J^T   (:234)
    SigS = symmetrise(Fm, i);
    store_rows(S.Sigmas_smooth + qt * 16, SigS, i);
    S.mus_smooth[qt * 4 + i] = mus;
    s = nx;
  }
}


// =====================================================================================================================
// backward: the adjoint of lgssm_bwd.h (same equations) in the quad layout, with the gains K | S | J saved by the forward.
// ws record per (b,t), 40 floats: [ g mu_f (4) | g Sig_f (16) | g mu_p (4) | g Sig_p (16) ], matrices by rows.
// =====================================================================================================================
constexpr int WS_REC = 2 * (4 + 16);

template <bool HAS_FP>
__device__ __forceinline__ void rts_bwd_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                                              const kvae_lgssm_input_grads &G, float *ws, int b, int i, int lane) {
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  float *w = ws + bT * WS_REC;
  float gsm = U.mus_smooth[bT * 4 + i];
  Mat gsS = load_rows(U.Sigmas_smooth + bT * 16, i);
  w[4 + 16 + i] = HAS_FP ? U.mus_pred[bT * 4 + i] : 0.0f;
  store_rows(w + 4 + 16 + 4, HAS_FP ? load_rows(U.Sigmas_pred + bT * 16, i) : zero(), i);
  store_rows(gstack_at(G.gA, b, 0), zero(), i);
  struct In {
    Mat Sf, Sp, Ss, A, J, uSs, uSf, uSp;
    float mup, mus, uMs, uMf, uMp;
  } s, nx;
  int64_t q = bT;
  const float *pA = stack_at(P.A, b, T >= 2 ? 1 : 0);
  const int64_t sA = P.A.st;
  auto load = [&](In &o) {
    o.Sf = load_rows(S.Sigmas_filt + q * 16, i);
    o.Sp = load_rows(S.Sigmas_pred + (q + 1) * 16, i);
    o.Ss = load_rows(S.Sigmas_smooth + (q + 1) * 16, i);
    o.A = load_rows(pA, i);
    o.J = load_rows(S.aux + q * KV_AUX_N4 + 12, i);
    o.mup = S.mus_pred[(q + 1) * 4 + i];
    o.mus = S.mus_smooth[(q + 1) * 4 + i];
    o.uMs = U.mus_smooth[(q + 1) * 4 + i];
    o.uSs = load_rows(U.Sigmas_smooth + (q + 1) * 16, i);
    if constexpr (HAS_FP) {
      o.uMf = U.mus_filt[q * 4 + i];
      o.uSf = load_rows(U.Sigmas_filt + q * 16, i);
      o.uMp = U.mus_pred[(q + 1) * 4 + i];
      o.uSp = load_rows(U.Sigmas_pred + (q + 1) * 16, i);
    }
  };
  if (T >= 2) load(s);
  nx = s;
  KV_Q4_DRAIN();
  for (int t = 0; t + 1 < T; ++t) {
    if (t + 2 < T) q += 1, pA += sA;
    load(nx);
    KV_Q4_FENCE();
    float *wt = w + (int64_t)t * WS_REC;
    Mat gM = symmetrise(gsS, i);
    const Mat Spt = transpose(s.Sp, i);
    Mat D2 = add(sub(s.Ss, s.Sp), sub(s.Ss, Spt));              // D + D^T, D = Sig_s[t+1] - Sig_p[t+1]
    Mat Jt = transpose(s.J, i);
    float dmu = s.mus - s.mup;
    guard(gsm, dmu);
    guard(D2);
    const Mat Y1 = mul_nn(gM, s.J);                              // gM J
    Mat gJ = mul_nn(Y1, D2);                                     // Y1 (D^T + D)
    outer_acc(gJ, gsm, dmu);                                     //   + gsm dmu^T
    const Mat gD = mul_nn(Jt, Y1);                               // J^T Y1
    const float gdm = matvec(Jt, gsm, 0.0f);                     // J^T gsm
    const Mat gJt = transpose(gJ, i);
    Mat gR = solve(s.Sp, gJt, i, lane);                          // Sig_p gR = gJ^T
    Mat gRt = transpose(gR, i);
    guard(gR);
    guard(gRt);
    const Mat gWA = mul_nn(gRt, s.A);                            // gR^T A[t+1]
    const Mat gP = mul_nn(Jt, gRt);                              // J^T gR^T
    const Mat gAs = mul_nn(gR, s.Sf);                            // gR Sig_f: the smoother's share of gA[t+1]
    store_rows(wt + 4, add(add(HAS_FP ? s.uSf : zero(), gM), gWA), i);
    store_rows(wt + WS_REC + 4 + 16 + 4, sub(sub(HAS_FP ? s.uSp : zero(), gD), gP), i);
    store_rows(gstack_at(G.gA, b, t + 1), gAs, i);
    wt[i] = (HAS_FP ? s.uMf : 0.0f) + gsm;
    wt[WS_REC + 4 + 16 + i] = (HAS_FP ? s.uMp : 0.0f) - gdm;
    gsS = add(s.uSs, gD);
    gsm = s.uMs + gdm;
    s = nx;
  }
  float *wl = w + (int64_t)(T - 1) * WS_REC;
  const int64_t ql = bT + T - 1;
  store_rows(wl + 4, add(HAS_FP ? load_rows(U.Sigmas_filt + ql * 16, i) : zero(), gsS), i);
  wl[i] = (HAS_FP ? U.mus_filt[ql * 4 + i] : 0.0f) + gsm;
}

template <bool HAS_GQ>
__device__ __forceinline__ void filter_bwd_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                                                 const kvae_lgssm_input_grads &G, const float *ws, int b, int i, int lane) {
  const int T = P.T;
  const int64_t bT = (int64_t)b * T;
  const float R00 = P.R[0], R01 = P.R[1], R10 = P.R[2], R11 = P.R[3];
  const Mat I4 = eye(i);
  struct In {
    Mat A, Bm, Sig, Sp, wSf, wSp, gAs;
    float Cl0, Cl1, u, mk, mu, mup, wmf, wmp;
    f2 y, ku;
    f4 Sv;
  } s, nx;
  StepPtr ptr;
  ptr.init(P, b, T - 1);
  int t_ld = T - 1;
  auto load = [&](In &o) {
    const int64_t q = bT + t_ld;
    o.A = load_rows(ptr.A, i), o.Bm = load_rows(ptr.Bm, i);
    o.Cl0 = ptr.C[i], o.Cl1 = ptr.C[4 + i];
    o.u = ptr.U[i];
    o.y = *reinterpret_cast<const f2 *>(ptr.Y);
    o.mk = *ptr.mk;
    const float *pS = t_ld > 0 ? S.Sigmas_filt + (q - 1) * 16 : P.Sigma0 + (int64_t)b * P.Sigma0_sb;
    const float *pm = t_ld > 0 ? S.mus_filt + (q - 1) * 4 : P.mu0 + (int64_t)b * P.mu0_sb;
    o.Sig = load_rows(pS, i);
    o.mu = pm[i];
    o.Sp = load_rows(S.Sigmas_pred + q * 16, i);
    o.mup = S.mus_pred[q * 4 + i];
    const float *ax = S.aux + q * KV_AUX_N4;
    o.ku = *reinterpret_cast<const f2 *>(ax + 2 * i);
    o.Sv = *reinterpret_cast<const f4 *>(ax + 8);
    const float *w = ws + q * WS_REC;
    o.wmf = w[i];
    o.wSf = load_rows(w + 4, i);
    o.wmp = w[4 + 16 + i];
    o.wSp = load_rows(w + 4 + 16 + 4, i);
    o.gAs = load_rows(gstack_at(G.gA, b, t_ld), i);
  };
  load(s);
  nx = s;
  KV_Q4_DRAIN();
  float gmu = 0.0f;
  Mat gSig = zero();
  for (int t = T - 1; t >= 0; --t) {
    if (t >= 1) t_ld = t - 1, ptr.step(-1);
    load(nx);
    KV_Q4_FENCE();
    const int64_t q = bT + t;
    const float mk = P.mask ? s.mk : 1.0f;
    gmu += s.wmf;
    gSig = add(gSig, s.wSf);
    Mat Gm = symmetrise(gSig, i);
    float k0 = mk * s.ku[0], k1 = mk * s.ku[1];
    Mat M = I4;
    const float nk0 = -k0, nk1 = -k1;
    outer_acc(M, nk0, s.Cl0);
    outer_acc(M, nk1, s.Cl1);
    const float gr0 = qsum(k0 * gmu), gr1 = qsum(k1 * gmu);          // gr = K^T gmu
    const float r0 = s.y[0] - qsum(s.Cl0 * s.mup), r1 = s.y[1] - qsum(s.Cl1 * s.mup);
    const Mat Spt = transpose(s.Sp, i);
    Mat Sp2 = add(s.Sp, Spt);
    Mat Mt = transpose(M, i);
    guard(Gm);
    guard(Sp2);
    guard(k0, k1);
    const Mat X1 = mul_nn(Gm, M);                                     // G (I - K C)
    const Mat gIKC = mul_nn(X1, Sp2);                                 // X1 (Sig_p^T + Sig_p)
    Mat gSp = mul_nn_acc(Mt, X1, s.wSp);                              // (I - K C)^T X1 + handed-off
    // gK = G K (R^T + R) - gIKC C^T + gmu r^T   (lane i: row i)
    const float GK0 = matvec(Gm, k0, 0.0f), GK1 = matvec(Gm, k1, 0.0f);
    const float gK0 = GK0 * (R00 + R00) + GK1 * (R01 + R10) - matvec(gIKC, s.Cl0, 0.0f) + gmu * r0;
    const float gK1 = GK0 * (R10 + R01) + GK1 * (R11 + R11) - matvec(gIKC, s.Cl1, 0.0f) + gmu * r1;
    // Z = S^{-T} (mask gK^T): column i on lane i
    const Inv2 F = factor2(s.Sv[0], s.Sv[1], s.Sv[3]);
    float z0, z1;
    solve2(F, mk * gK0, mk * gK1, z0, z1);
    const float zk00 = qsum(z0 * s.ku[0]), zk01 = qsum(z0 * s.ku[1]), zk10 = qsum(z1 * s.ku[0]), zk11 = qsum(z1 * s.ku[1]);
    const float h00 = -0.5f * (zk00 + zk00), h01 = -0.5f * (zk01 + zk10), h11 = -0.5f * (zk11 + zk11);
    float gCP0 = h00 * s.Cl0 + h01 * s.Cl1, gCP1 = h01 * s.Cl0 + h11 * s.Cl1;   // gCP = gS0 C (lane j: column j)
    // gSp += Z^T C + C^T gCP
    guard(gCP0, gCP1);
    outer_acc(gSp, z0, s.Cl0);
    outer_acc(gSp, z1, s.Cl1);
    outer_acc(gSp, s.Cl0, gCP0);
    outer_acc(gSp, s.Cl1, gCP1);
    // gC = -K^T gIKC + Z Sig_p + gS0 (C Sig_p) + gCP Sig_p^T - gr mu_p^T    (lane j: column j)
    Mat gIKCt = transpose(gIKC, i);
    guard(z0, z1);
    const float cp0 = matvec(Spt, s.Cl0, 0.0f), cp1 = matvec(Spt, s.Cl1, 0.0f);
    const float gC0 = -matvec(gIKCt, k0, 0.0f) + matvec(Spt, z0, 0.0f) + (h00 * cp0 + h01 * cp1) + matvec(s.Sp, gCP0, 0.0f) - gr0 * s.mup;
    const float gC1 = -matvec(gIKCt, k1, 0.0f) + matvec(Spt, z1, 0.0f) + (h01 * cp0 + h11 * cp1) + matvec(s.Sp, gCP1, 0.0f) - gr1 * s.mup;
    float *gCo = gstack_at(G.gC, b, t);
    gCo[i] = gC0, gCo[4 + i] = gC1;
    float gmp = gmu + s.wmp - (s.Cl0 * gr0 + s.Cl1 * gr1);           // gmp = gmu + handed-off - C^T gr
    G.gY[q * 2 + (i & 1)] = (i & 1) ? gr1 : gr0;
    if constexpr (HAS_GQ) store_rows(gstack_at(G.gQ, b, t), gSp, i);
    // gA[t] = smoother share + gSp^T (A Sig) + (gSp A) Sig^T + gmp mu^T ; carried adjoints for t-1
    Mat gSpt = transpose(gSp, i);
    Mat At = transpose(s.A, i), Bt = transpose(s.Bm, i);
    guard(gmp);
    const Mat AS = mul_nn(s.A, s.Sig);
    const Mat gAS = mul_nn(gSp, s.A);
    Mat gA = s.gAs;
    outer_acc(gA, gmp, s.mu);
    gA = mul_nt(gAS, s.Sig, gA);
    gA = mul_nn_acc(gSpt, AS, gA);
    store_rows(gstack_at(G.gA, b, t), gA, i);
    gSig = mul_nn(At, gAS);                                           // A^T (gSp A)
    gmu = matvec(At, gmp, 0.0f);                                      // A^T gmp
    Mat gB = zero();
    outer_acc(gB, gmp, s.u);
    store_rows(gstack_at(G.gB, b, t), gB, i);
    G.gU[q * 4 + i] = matvec(Bt, gmp, 0.0f);                          // B^T gmp
    s = nx;
  }
  if (G.g_mu0) G.g_mu0[(int64_t)b * 4 + i] = gmu;
  if (G.g_Sigma0) store_rows(G.g_Sigma0 + (int64_t)b * 16, gSig, i);
}

}  // namespace q4
}  // namespace kvae
#endif
