// lgssm_q4.h — the QUAD LAYOUT of the (n, m, p) = (4, 4, 2) kernels: sixteen sequences per wavefront, a sequence owns one quad
// of lanes, lane i of the quad owns ROW i of every 4x4 matrix (4 VGPRs per matrix) and element i of every vector.  This is the
// rows-on-lanes layout of lgssm_n16.h shrunk to quads.  This file holds what the layout needs besides products: quad
// broadcasts / reductions (DPP quad_perm), 16-byte row loads and stores, the per-lane 2x2 innovation solve, the running operand
// pointers, the prefetch fences, and the PIVOTED 4x4 solve (Gauss-Jordan with partial pivoting per quad: same pivot sequence and
// multipliers as torch.linalg.solve / getrf; pivot row by quad broadcast folded into the FMA - v_fmac_f32_dpp, spelled as inline
// asm because hipcc does not fold a DPP mov into an fma; the row exchange is a ds_bpermute, taken only when some quad needs
// one).  No LDS, no barrier.
//
// The sweeps themselves - filter, RTS smoother, their adjoints - are in lgssm_m4.h, with the 4x4 products on the matrix cores.
// (Rounds 2-3 kept a second set of sweeps here whose products were 16 DPP-folded FMAs and whose transposes were quad_perm
// butterflies; the matrix-core sweeps beat them at every batch size - 55 / 89 us against 83 / 129 us forward / backward at 256
// sequences of T = 50, 425 / 779 against 489 / 825 us at 32768 - and they were removed.)
//
// The one-wavefront-per-sequence kernels of lgssm_n4.h remain for operands that are not 16-byte aligned and as the reference the
// tests compare the quad kernels with.  Same equations, cited in lgssm_fwd.h / lgssm_bwd.h; same aux record (K unmasked | S | J).
#pragma once
#include "lgssm_n4.h"   // stack_at, mask_addr, KV_AUX_N4

#if (!defined(KVAE_HOSTSIM) || defined(KVAE_WAVE_EMU)) && !defined(KV_TPP)   // KVAE_WAVE_EMU: tests/hostsim/wave_emu.h
namespace kvae {
namespace q4 {

using f4 = __attribute__((ext_vector_type(4))) float;
using f2 = __attribute__((ext_vector_type(2))) float;
struct Mat { float c[4]; };   // lane i: row i

// acc += (quad lane K of src) * f
template <int K> __device__ __forceinline__ void fmac_q(float &acc, float src, float f);
#if defined(KVAE_WAVE_EMU)
template <int K> __device__ __forceinline__ void fmac_q(float &acc, float src, float f) {
  acc = fmaf(__builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), K * 0x55, 0xf, 0xf, true)), f, acc);
}
#else
#define KV_Q4_DPP(K)                                                                                                        \
  template <> __device__ __forceinline__ void fmac_q<K>(float &acc, float src, float f) {                                  \
    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #K "," #K "," #K "," #K "] row_mask:0xf bank_mask:0xf"            \
                 : "+v"(acc) : "v"(src), "v"(f));                                                                          \
  }
KV_Q4_DPP(0) KV_Q4_DPP(1) KV_Q4_DPP(2) KV_Q4_DPP(3)
#undef KV_Q4_DPP
#endif

// Two wait states between a VALU write and a DPP read of the same register: the compiler's hazard recogniser cannot see
// into the asm above, so every value that the compiler's own VALU code produced and that is about to be read THROUGH DPP
// passes through one of these first.  The "+v" operands tie the s_nop into the data flow: the producer cannot be scheduled
// after it, the consumer not before it.  (Values produced by the asm itself are ordered by `volatile`.)

template <int CTRL>
__device__ __forceinline__ float dppm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int K> __device__ __forceinline__ float qb(float v) { return dppm<K * 0x55>(v); }      // quad broadcast of lane K
__device__ __forceinline__ float qx1(float v) { return dppm<0xB1>(v); }                          // lane ^ 1
__device__ __forceinline__ float qx2(float v) { return dppm<0x4E>(v); }                          // lane ^ 2
__device__ __forceinline__ float qsum(float x) { x += qx1(x); x += qx2(x); return x; }           // over the quad, all lanes
__device__ __forceinline__ float qmax(float x) { x = fmaxf(x, qx1(x)); x = fmaxf(x, qx2(x)); return x; }

#if defined(KVAE_WAVE_EMU)
__device__ __forceinline__ void guard(Mat &) {}
#else
__device__ __forceinline__ void guard(Mat &m) { asm volatile("s_nop 1" : "+v"(m.c[0]), "+v"(m.c[1]), "+v"(m.c[2]), "+v"(m.c[3])); }
#endif

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
  guard(m);     // rows may come from the compiler's own code (the caller's, the exchange above)
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
#define KV_Q4_FENCE() asm volatile("" ::: "memory")
#define KV_Q4_DRAIN()                      \
  do {                                     \
    asm volatile("" ::: "memory");         \
    __builtin_amdgcn_s_waitcnt(0x0F70);    \
    asm volatile("" ::: "memory");         \
  } while (0)

// ws record of the backward per (b,t), 40 floats: [ g mu_f (4) | g Sig_f (16) | g mu_p (4) | g Sig_p (16) ], matrices by rows
constexpr int WS_REC = 2 * (4 + 16);

}  // namespace q4
}  // namespace kvae
#endif
