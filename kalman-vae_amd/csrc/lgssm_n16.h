// lgssm_n16.h — filter / RTS smoother / their adjoints for (n, m, p) = (16, 16, 2) on the f32 matrix cores.
//
// One 64-lane wavefront per sequence, the whole T loop in-kernel, every 16x16 matrix in FOUR VGPRs:
//
//   C-layout of X :  lane l = (j = l & 15, g = l >> 4), register r  holds  X[4g + r][j]
//
// which is exactly the accumulator layout of v_mfma_f32_16x16x4_f32 AND a legal A- or B-operand layout for it
// (operand register r of lane (x, g) is read as A[x][k=g] / B[k=g][x]; the k order inside a sum is free as long as A
// and B agree).  Feeding two C-layout matrices M, N as A and B, four MFMAs give
//
//   P(M, N) = M^T N        (in C-layout again; an exact k-ordered fp32 fma chain, so parity is unaffected)
//
// so a congruence X S X^T is P(P(S, Xt), Xt) with Xt = C-layout of X^T, in the reference's own association order
// ((X S) X^T, kalman_filter.py:67, :99, :234).  Xt is what a 16-byte row load of a row-major X delivers (A_t), what
// I - K C is built as directly, and one LDS transpose away for the smoother gain J.  Nothing else goes through LDS:
// vectors live on lanes (L-layout: lane j holds v[j]) or on registers (W-layout: register r of row-group g holds
// v[4g + r]); M^T v maps W -> L with 4 fmas + a cross-row-group sum (v_permlane16/32_swap), M v maps L -> W with
// 4 multiplies + DPP row reductions.
//
// The n x n smoother system (kalman_filter.py:229, torch.linalg.solve = LU with partial pivoting) is a Gauss-Jordan
// elimination with partial pivoting held entirely in registers: lane i owns row i of the matrix (16 registers,
// replicated in the four row-groups) and row i of 4 of the 16 right-hand sides; the pivot row reaches every lane through
// DPP row_newbcast folded into the fma, row exchanges (only when the pivot is not already on the diagonal) are
// ds_bpermutes, and the result leaves in C-layout of J.  Same pivot sequence and multipliers as getrf.
#pragma once
#include "lgssm_n4.h"   // solve2, stack_at, mask_addr

#if (!defined(KVAE_HOSTSIM) || defined(KVAE_WAVE_EMU)) && !defined(KV_TPP)   // KVAE_WAVE_EMU: tests/hostsim/wave_emu.h
namespace kvae {
namespace n16 {

using f4 = __attribute__((ext_vector_type(4))) float;
constexpr int N = 16, NN = 256, P2 = 2;
constexpr int LD = 20;                       // leading dimension of the LDS tiles: conflict-free b128 rows, 16-byte aligned
#define KV_AUX_N16 (N * P2 + P2 * P2 + NN)   // K unmasked [16,2] | S [2,2] | J [16,16], as KVAE_AUX(16, 2)

struct alignas(16) Lds { float t[N * LD]; };

__device__ __forceinline__ unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float bitsf(unsigned x) { return __builtin_bit_cast(float, x); }

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// value of lane K of the caller's 16-lane row, in every lane of that row
template <int K>
__device__ __forceinline__ float bcast(float v) { return dpp<0x150 + K>(v); }
// sum / max over the 16 lanes of a row (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror): every lane gets the
// same bits (each step adds the same two partial sums in both partners)
__device__ __forceinline__ float row_sum(float x) {
  x += dpp<0xB1>(x);
  x += dpp<0x4E>(x);
  x += dpp<0x141>(x);
  x += dpp<0x140>(x);
  return x;
}
__device__ __forceinline__ float row_max(float x) {
  x = fmaxf(x, dpp<0xB1>(x));
  x = fmaxf(x, dpp<0x4E>(x));
  x = fmaxf(x, dpp<0x141>(x));
  x = fmaxf(x, dpp<0x140>(x));
  return x;
}
// sum over the four row-groups (lanes j, j+16, j+32, j+48), result in all four
__device__ __forceinline__ float xg_sum(float x) {
  auto a = __builtin_amdgcn_permlane16_swap(fbits(x), fbits(x), false, false);
  const float s = bitsf(a[0]) + bitsf(a[1]);
  auto b = __builtin_amdgcn_permlane32_swap(fbits(s), fbits(s), false, false);
  return bitsf(b[0]) + bitsf(b[1]);
}
__device__ __forceinline__ float dot4(f4 a, f4 b) { return fmaf(a[3], b[3], fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]))); }

// acc + M^T N
__device__ __forceinline__ f4 mtn(f4 M, f4 Nm, f4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(M[0], Nm[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(M[1], Nm[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(M[2], Nm[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(M[3], Nm[3], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f4 mtn(f4 M, f4 Nm) { return mtn(M, Nm, f4{0.f, 0.f, 0.f, 0.f}); }

// (M^T v)[j] for v in W-layout -> L-layout
__device__ __forceinline__ float mtv(f4 M, f4 vW) { return xg_sum(dot4(M, vW)); }
// (M v)[4g + r] for v in L-layout -> W-layout
__device__ __forceinline__ f4 mv(f4 M, float vL) {
  return f4{row_sum(M[0] * vL), row_sum(M[1] * vL), row_sum(M[2] * vL), row_sum(M[3] * vL)};
}
// L-layout -> W-layout (any lane of row-group g with j = 4g + r holds the element)
__device__ __forceinline__ f4 l2w(float vL, int lane) {
  const int base = (lane & 48) | ((lane >> 4) << 2);
  return f4{__shfl(vL, base, 64), __shfl(vL, base + 1, 64), __shfl(vL, base + 2, 64), __shfl(vL, base + 3, 64)};
}

// C-layout of X^T from C-layout of X through an LDS tile
__device__ __forceinline__ f4 transpose(f4 X, Lds &L, int j, int g) {
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) L.t[(4 * g + r) * LD + j] = X[r];
  __syncthreads();
  return *reinterpret_cast<const f4 *>(&L.t[j * LD + 4 * g]);
}
// (The same two as X^T = X^T I on the matrix cores - four dependent 16x16x4 MFMAs, no LDS - were measured in the adjoint sweeps:
// 752 -> 778 us at the configs[4] shard.  An LDS round trip costs fewer cycles of the wavefront than a 128-cycle MFMA chain.)
__device__ __forceinline__ f4 symmetrise(f4 X, Lds &L, int j, int g) {
  const f4 Xt = transpose(X, L, j, g);
  return f4{0.5f * (X[0] + Xt[0]), 0.5f * (X[1] + Xt[1]), 0.5f * (X[2] + Xt[2]), 0.5f * (X[3] + Xt[3])};
}

// ---- global-memory access in the three layouts -------------------------------------------------------------------
// C-layout of X^T == 16-byte loads of the rows of a row-major X (for a symmetric X this is X itself)
__device__ __forceinline__ f4 load_rows(const float *X, int j, int g) { return *reinterpret_cast<const f4 *>(X + j * N + 4 * g); }
__device__ __forceinline__ void store_rows(float *X, f4 v, int j, int g) { *reinterpret_cast<f4 *>(X + j * N + 4 * g) = v; }
// C-layout of X itself: four 64-byte row segments per register
__device__ __forceinline__ f4 load_c(const float *X, int j, int g) {
  return f4{X[(4 * g + 0) * N + j], X[(4 * g + 1) * N + j], X[(4 * g + 2) * N + j], X[(4 * g + 3) * N + j]};
}
__device__ __forceinline__ void store_c(float *X, f4 v, int j, int g) {
#pragma unroll
  for (int r = 0; r < 4; ++r) X[(4 * g + r) * N + j] = v[r];
}
__device__ __forceinline__ f4 load_w(const float *v, int g) { return *reinterpret_cast<const f4 *>(v + 4 * g); }
// every lane of a row-group stores the same 16 bytes (the coalescer merges them): NO branch around a store - the
// compiler's s_waitcnt vmcnt(N) for the prefetched operands may only count stores that are issued unconditionally, and
// every store it cannot count turns into a wait for an older store's completion
__device__ __forceinline__ void store_w(float *v, f4 x, int g) { *reinterpret_cast<f4 *>(v + 4 * g) = x; }

// ---- n x n solve: Gauss-Jordan in registers -------------------------------------------------------------------------
// Solve  Mat X = RHS  where lane i (= l & 15, replicated over the row-groups) holds row i of Mat in m[0..15] and row i of
// RHS columns 4g..4g+3 in x.  Returns row i of X (columns 4g..4g+3): with Mat = Sigma_pred^T and RHS = A Sigma_filt this is
// the C-layout of J (kalman_filter.py:229).
//
// The reference calls torch.linalg.solve (getrf: LU with partial pivoting).  A predicted covariance is symmetric positive
// definite whenever the model is sane, and for such a matrix elimination WITHOUT row exchanges is backward stable (growth
// factor 1), so the fast path eliminates in natural order: no pivot search, no wave-uniform branch, every broadcast a
// compile-time DPP row_newbcast.  It watches the pivots; a non-positive or non-finite one (learned Q that is not PSD - the
// reference's own stability recipe produces such matrices) sends the whole solve to the pivoted elimination below, which
// follows getrf's pivot sequence (first maximum of |column|) and multipliers.  (Watching the multipliers as well - partial
// pivoting keeps them <= 1 - was tried: a bound of 16 sends ordinary covariances of the configs[4] shard down the slow path,
// 289 -> 745 us for the smoother sweep, and does not move the error of the indefinite test case.)
__device__ __forceinline__ float frcp(float x) {   // 1/x: hardware reciprocal + one Newton step (<= 1 ulp)
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}
// acc += f * (lane K of acc's row), ONE instruction.  hipcc does not fold a DPP mov into v_fma (VOP3), so it is spelled out.
// DPP hazard (a VALU write of the DPP source needs 2 wait states before the read): the source here is always a row register
// last written by an earlier fmac of the previous elimination step, by an LDS return, or by the s_nop-guarded head of the step.
template <int K>
__device__ __forceinline__ void fmac_bcast(float &acc, float f) {
#if defined(KVAE_WAVE_EMU)
  acc = fmaf(bcast<K>(acc), f, acc);
#else
  asm volatile("v_fmac_f32_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(f), "n"(K));
#endif
}
template <int K, int C>
__device__ __forceinline__ void gj_update_cols(float (&m)[N], float f) {
  if constexpr (C < N) {
    fmac_bcast<K>(m[C], f);
    gj_update_cols<K, C + 1>(m, f);
  }
}
template <int K>
__device__ __forceinline__ void gj_eliminate(float (&m)[N], f4 &x, float &my_rinv, int i) {
  const float piv = bcast<K>(m[K]);
  const float rinv = frcp(piv);
  const float f = i == K ? 0.0f : -(m[K] * rinv);
#if !defined(KVAE_WAVE_EMU)
  asm volatile("s_nop 1");                     // whatever the compiler scheduled last, the DPP reads below are clear of it
#endif
  gj_update_cols<K, K + 1>(m, f);
#pragma unroll
  for (int q = 0; q < 4; ++q) x[q] = fmaf(f, bcast<K>(x[q]), x[q]);
  if (i == K) my_rinv = rinv;
}
// natural-order elimination; `bad` collects pivots that are not > 0
template <int K>
__device__ __forceinline__ void gj_step_spd(float (&m)[N], f4 &x, float &my_rinv, bool &bad, int i) {
  bad |= !(bcast<K>(m[K]) > 0.0f);
  gj_eliminate<K>(m, x, my_rinv, i);
  if constexpr (K + 1 < N) gj_step_spd<K + 1>(m, x, my_rinv, bad, i);
}
// partial pivoting: rows K..15 compete for column K (first maximum wins, as in getrf); rows < K are spent pivots
template <int K>
__device__ __forceinline__ void gj_step_piv(float (&m)[N], f4 &x, float &my_rinv, int i, int lane) {
  const float cand = i >= K ? fabsf(m[K]) : -1.0f;
  const float mx = row_max(cand);
  const unsigned long long hit = __ballot(cand == mx) & 0xffffull;
  const int p = hit ? (int)__builtin_ctzll(hit) : K;          // all-NaN column: keep the diagonal, NaNs propagate
  if (p != K) {                                                // wave-uniform: exchange rows K and p
    const int src = (lane & 48) | (i == K ? p : (i == p ? K : i));
#pragma unroll
    for (int c = K; c < N; ++c) m[c] = __shfl(m[c], src, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = __shfl(x[q], src, 64);
  }
  gj_eliminate<K>(m, x, my_rinv, i);
  if constexpr (K + 1 < N) gj_step_piv<K + 1>(m, x, my_rinv, i, lane);
}
// lane i <- row i of the tile (the 16 entries of row i of Mat)
__device__ __forceinline__ void read_rows(float (&m)[N], const Lds &L, int j) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f4 v = *reinterpret_cast<const f4 *>(&L.t[j * LD + 4 * q]);
    m[4 * q] = v[0], m[4 * q + 1] = v[1], m[4 * q + 2] = v[2], m[4 * q + 3] = v[3];
  }
}
// X = Mat^{-1} RHS with row j of Mat in m0 (which stays intact for the pivoted repeat) and the rows of RHS on the lanes
__device__ __forceinline__ f4 solve_rows(const float (&m0)[N], f4 rhs, int lane) {
  const int j = lane & 15;
  float m[N];
#pragma unroll
  for (int c = 0; c < N; ++c) m[c] = m0[c];
  f4 x = rhs;
  float my_rinv = 0.0f;
  bool bad = false;
  gj_step_spd<0>(m, x, my_rinv, bad, j);
  if (__builtin_expect(__any(bad), 0)) {                       // not positive definite: redo with row exchanges
#pragma unroll
    for (int c = 0; c < N; ++c) m[c] = m0[c];
    x = rhs;
    gj_step_piv<0>(m, x, my_rinv, j, lane);
  }
  return f4{x[0] * my_rinv, x[1] * my_rinv, x[2] * my_rinv, x[3] * my_rinv};
}
// X = (Xc^T)^{-1} RHS with Xc in C-layout (so Mat = Xc^T has its rows on the lanes after one LDS hop) and RHS rows on lanes
__device__ __forceinline__ f4 solve_transposed(f4 Xc, f4 rhs, Lds &L, int lane) {
  const int j = lane & 15, g = lane >> 4;
  __syncthreads();
  *reinterpret_cast<f4 *>(&L.t[j * LD + 4 * g]) = Xc;          // t[a][b] = Xc[b][a] = Mat[a][b]
  __syncthreads();
  float m[N];
  read_rows(m, L, j);
  return solve_rows(m, rhs, lane);
}
// row j of a row-major 16 x 16 matrix, whole (the four row-groups read the same 64 bytes)
struct Row16 { f4 q[4]; };
__device__ __forceinline__ Row16 load_row16(const float *X, int j) {
  const f4 *r = reinterpret_cast<const f4 *>(X + j * N);
  return Row16{{r[0], r[1], r[2], r[3]}};
}

// ---- per-step operands ---------------------------------------------------------------------------------------------
struct StepIn {
  f4 At, Bt, Qc, Cw0, Cw1, uW;
  float Cl0, Cl1, y0, y1, mk;
};
// Running pointers to the operands of one step of one sequence; advance() moves them by one time step (strides in floats).
struct StepPtr {
  const float *A, *Bm, *Q, *C, *U, *Y, *mk;
  int64_t sA, sB, sQ, sC, smk;
  __device__ __forceinline__ void init(const kvae_lgssm_problem &P, int b) {
    A = stack_at(P.A, b, 0), Bm = stack_at(P.Bm, b, 0), Q = stack_at(P.Q, b, 0), C = stack_at(P.C, b, 0);
    sA = P.A.st, sB = P.Bm.st, sQ = P.Q.st, sC = P.C.st;
    U = P.U + (int64_t)b * P.T * N, Y = P.Y + (int64_t)b * P.T * P2;
    mk = mask_addr(P, b, 0), smk = P.mask ? 1 : 0;
  }
  __device__ __forceinline__ void advance() { A += sA, Bm += sB, Q += sQ, C += sC, U += N, Y += P2, mk += smk; }
};
__device__ __forceinline__ void load_step(const StepPtr &p, int j, int g, StepIn &s) {
  s.At = load_rows(p.A, j, g);
  s.Bt = load_rows(p.Bm, j, g);
  s.Qc = load_c(p.Q, j, g);
  s.Cw0 = load_w(p.C, g), s.Cw1 = load_w(p.C + N, g);
  s.Cl0 = p.C[j], s.Cl1 = p.C[N + j];
  s.uW = load_w(p.U, g);
  s.y0 = p.Y[0], s.y1 = p.Y[1];
  s.mk = *p.mk;   // raw: the NULL-mask select happens at the point of use, or this load would be waited for (and with it,
                  // in order, every prefetch issued before it) as soon as it is issued
}
// Pins the prefetch in program order: the loads above it are issued before any store below it, so that the counted
// s_waitcnt vmcnt(N) that finally consumes them only has to see the loads done, not the step's stores (hipcc otherwise
// sinks the loads towards their first use, i.e. behind this step's stores).
#define KV_PREFETCH_FENCE() asm volatile("" ::: "memory")
// vmcnt(0) that stays where it is written: the builtin (which the s_waitcnt pass understands, unlike inline asm) is not
// ordered against plain loads by itself, so it is bracketed by compiler-level memory fences.
#define KV_DRAIN_VMEM()                      \
  do {                                       \
    asm volatile("" ::: "memory");           \
    __builtin_amdgcn_s_waitcnt(0x0F70);      \
    asm volatile("" ::: "memory");           \
  } while (0)

// x = S^{-1} b for the symmetric 2x2 innovation covariance (partial pivoting, as lu_solve / getrf would do it)
struct Inv2 { float a00, a01, r00, l, ru11; bool sw; };
__device__ __forceinline__ Inv2 factor2(float s00, float s01, float s11) {
  Inv2 o;
  o.sw = fabsf(s01) > fabsf(s00);
  o.a00 = o.sw ? s01 : s00, o.a01 = o.sw ? s11 : s01;
  const float a10 = o.sw ? s00 : s01, a11 = o.sw ? s01 : s11;
  o.r00 = frcp(o.a00);
  o.l = a10 * o.r00;
  o.ru11 = frcp(fmaf(-o.l, o.a01, a11));
  return o;
}
__device__ __forceinline__ Sol2 solve2f(const Inv2 &F, float b0, float b1) {
  const float c0 = F.sw ? b1 : b0, c1 = F.sw ? b0 : b1;
  Sol2 o;
  o.x1 = fmaf(-F.l, c0, c1) * F.ru11;
  o.x0 = fmaf(-F.a01, o.x1, c0) * F.r00;
  return o;
}

// Innovation statistics shared by the forward and the backward sweep.
struct Gain {
  float cp0, cp1;        // (C Sigp)[c][j]               L-layout
  f4 pct0, pct1;         // (Sigp C^T)[4g+r][c]          W-layout
  float s00, s01, s11;   // S = sym(C Sigp C^T + R)
  float kl0, kl1;        // unmasked gain K[j][c]        L-layout
  f4 kw0, kw1;           // unmasked gain K[4g+r][c]     W-layout
};
__device__ __forceinline__ void innovation(const StepIn &s, f4 Sigp, const float *R, Gain &G) {
  G.cp0 = mtv(Sigp, s.Cw0), G.cp1 = mtv(Sigp, s.Cw1);
  G.pct0 = mv(Sigp, s.Cl0), G.pct1 = mv(Sigp, s.Cl1);
  const float a00 = row_sum(G.cp0 * s.Cl0) + R[0], a01 = row_sum(G.cp0 * s.Cl1) + R[1];
  const float a10 = row_sum(G.cp1 * s.Cl0) + R[2], a11 = row_sum(G.cp1 * s.Cl1) + R[3];
  G.s00 = 0.5f * (a00 + a00), G.s01 = 0.5f * (a01 + a10), G.s11 = 0.5f * (a11 + a11);   // kalman_filter.py:79
  const Inv2 F = factor2(G.s00, G.s01, G.s11);
  const Sol2 kl = solve2f(F, G.cp0, G.cp1);                                                // K^T = S^{-1} (Sigp C^T)^T  (:82-90)
  G.kl0 = kl.x0, G.kl1 = kl.x1;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const Sol2 kw = solve2f(F, G.pct0[r], G.pct1[r]);
    G.kw0[r] = kw.x0, G.kw1[r] = kw.x1;
  }
}

// ---- forward sweeps ------------------------------------------------------------------------------------------------
template <bool AUX>
__device__ __forceinline__ void filter_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, Lds &L) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, T = P.T;
  const int64_t bT = (int64_t)b * T;
  f4 Sig = load_c(P.Sigma0 + (int64_t)b * P.Sigma0_sb, j, g);      // C-layout of the prior itself: no symmetry assumed
  f4 muW = load_w(P.mu0 + (int64_t)b * P.mu0_sb, g);
  const float R[4] = {P.R[0], P.R[1], P.R[2], P.R[3]};
  StepIn s0, s1, s2;
  StepPtr ptr;
  ptr.init(P, b);
  load_step(ptr, j, g, s0);
  if (T > 1) ptr.advance();
  load_step(ptr, j, g, s1);
  s2 = s1;
  // Drain the preamble's loads HERE.  The s_waitcnt pass is not path-sensitive: loads still pending at loop entry become a
  // vmcnt(0) at the loop top, which on every later iteration waits for the previous step's STORES (vmcnt counts both).
  KV_DRAIN_VMEM();
  // One step.  The operands of step t + 2 fly while steps t and t + 1 compute: an HBM miss costs about one step of this sweep,
  // and a register copy of a prefetched value waits for its load, so the three operand sets rotate by NAME (the caller is
  // unrolled three times) instead of being copied.  Unconditional (the last iterations re-read the last step): a branch around
  // the prefetch makes the s_waitcnt pass merge two timelines and wait for this step's stores at the loop top.
  auto step = [&](int t, const StepIn &s, StepIn &far) {
    if (t + 2 < T) ptr.advance();
    load_step(ptr, j, g, far);
    KV_PREFETCH_FENCE();
    const int64_t q = bT + t;
    // predict (kalman_filter.py:65-67): (A Sig) A^T + Q in the reference's association order
    const float mupL = xg_sum(dot4(s.At, muW) + dot4(s.Bt, s.uW));
    const f4 E1 = mtn(Sig, s.At);                                   // Sig^T A^T = (A Sig)^T
    const f4 Sigp = mtn(E1, s.At, s.Qc);                            // (A Sig) A^T + Q
    store_c(S.Sigmas_pred + q * NN, Sigp, j, g);
    Gain G;
    innovation(s, Sigp, R, G);
    const float r0 = s.y0 - row_sum(s.Cl0 * mupL), r1 = s.y1 - row_sum(s.Cl1 * mupL);   // :73
    if constexpr (AUX) {  // gains for the backward: K unmasked | S
      float *ax = S.aux + q * KV_AUX_N16;
      *reinterpret_cast<f4 *>(ax + 8 * g) = f4{G.kw0[0], G.kw1[0], G.kw0[1], G.kw1[1]};
      *reinterpret_cast<f4 *>(ax + 8 * g + 4) = f4{G.kw0[2], G.kw1[2], G.kw0[3], G.kw1[3]};
      *reinterpret_cast<f4 *>(ax + N * P2) = f4{G.s00, G.s01, G.s01, G.s11};
    }
    const float mk = P.mask ? s.mk : 1.0f;
    const float kl0 = mk * G.kl0, kl1 = mk * G.kl1;                 // :92
    const f4 kw0 = mk * G.kw0, kw1 = mk * G.kw1;
    const f4 mupW = l2w(mupL, lane);
    const f4 mufW = mupW + kw0 * r0 + kw1 * r1;                     // :96
    store_w(S.mus_pred + q * N, mupW, g);
    store_w(S.mus_filt + q * N, mufW, g);
    // Joseph update (:97-101): Mt = C-layout of (I - K C)^T, (M Sigp) M^T + K R K^T
    f4 Mt, krk;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      Mt[r] = (j == 4 * g + r ? 1.0f : 0.0f) - (kl0 * s.Cw0[r] + kl1 * s.Cw1[r]);
      const float kr0 = kw0[r] * R[0] + kw1[r] * R[2], kr1 = kw0[r] * R[1] + kw1[r] * R[3];
      krk[r] = kr0 * kl0 + kr1 * kl1;
    }
    const f4 E2 = mtn(Sigp, Mt);                                    // Sigp^T M^T = (M Sigp)^T
    const f4 F0 = mtn(E2, Mt, krk);
    Sig = symmetrise(F0, L, j, g);
    store_rows(S.Sigmas_filt + q * NN, Sig, j, g);                  // symmetric: rows == columns
    muW = mufW;
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

template <bool AUX>
__device__ __forceinline__ void rts_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int b, Lds &L) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, T = P.T;
  const int64_t bT = (int64_t)b * T;
  // last step: smoothed = filtered (kalman_filter.py:251-256)
  f4 SigS = load_rows(S.Sigmas_filt + (bT + T - 1) * NN, j, g);
  float musL = S.mus_filt[(bT + T - 1) * N + j];
  store_rows(S.Sigmas_smooth + (bT + T - 1) * NN, SigS, j, g);
  S.mus_smooth[(bT + T - 1) * N + j] = musL;
  struct In { f4 Sf, Spc, At; float mufL, mupL; } s0, s1, s2;
  // running pointers of step t: Sigma_f[t], mu_f[t], Sigma_p[t+1], mu_p[t+1], A[t+1]
  const float *pSf = S.Sigmas_filt + (bT + T - 2) * NN, *pSp = S.Sigmas_pred + (bT + T - 1) * NN;
  const float *pmf = S.mus_filt + (bT + T - 2) * N, *pmp = S.mus_pred + (bT + T - 1) * N;
  const float *pA = stack_at(P.A, b, T - 1);
  const int64_t sA = P.A.st;
  auto load = [&](In &o) {
    o.Sf = load_rows(pSf, j, g);
    o.Spc = load_c(pSp, j, g);
    o.At = load_rows(pA, j, g);
    o.mufL = pmf[j];
    o.mupL = pmp[j];
  };
  auto back = [&]() { pSf -= NN, pSp -= NN, pmf -= N, pmp -= N, pA -= sA; };
  if (T >= 2) load(s0);
  s1 = s0;
  if (T >= 3) {
    back();
    load(s1);
  }
  s2 = s1;
  KV_DRAIN_VMEM();                      // see filter_sweep
  auto step = [&](int t, const In &s, In &far) {                    // operands two steps ahead, rotated by name: see filter_sweep
    if (t >= 2) back();
    load(far);                                                      // unconditional, see filter_sweep
    KV_PREFETCH_FENCE();
    const int64_t q = bT + t;
    // J = Sig_f A^T Sigp^{-1}  <=>  Sigp^T J^T = A Sig_f  (kalman_filter.py:229)
    const f4 W = mtn(s.Sf, s.At);                                   // Sig_f A^T in C-layout == rows of (A Sig_f) on lanes
    const f4 Jc = solve_transposed(s.Spc, W, L, lane);
    const f4 Jt = transpose(Jc, L, j, g);
    if constexpr (AUX) store_rows(S.aux + q * KV_AUX_N16 + N * P2 + P2 * P2, Jt, j, g);   // J row-major
    const f4 D = SigS - s.Spc;
    const f4 E3 = mtn(D, Jt);                                       // D^T J^T = (J D)^T
    const f4 F = mtn(E3, Jt, s.Sf);                                 // Sig_f + (J D) J^T   (:234)
    SigS = symmetrise(F, L, j, g);
    store_rows(S.Sigmas_smooth + q * NN, SigS, j, g);
    const f4 dW = l2w(musL - s.mupL, lane);
    musL = s.mufL + mtv(Jt, dW);                                    // :232
    S.mus_smooth[q * N + j] = musL;   // the four row-groups store the same value: no branch around a store
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
// backward: the hand-derived adjoint of lgssm_bwd.h (same equations, cited there) on the matrix cores.
//
//   rts_bwd_sweep     t = 0..T-2   adjoint of the smoother; hands (g mu_f, g Sig_f, g mu_p, g Sig_p) to the filter sweep
//                                  through ws and parks the smoother's share of gA[t+1] in the gA output
//   filter_bwd_sweep  t = T-1..0   adjoint of the filter; writes gA, gB, gC, gQ, gY, gU (+ g mu0, g Sigma0)
//
// The gains K (unmasked), S and J come from the forward (states.aux), so the only n x n solve left per step is the
// adjoint of the smoother gain, Sig_p gR = gJ^T.  A product that the generic body writes with a transposed factor is a
// second MFMA chain here (X^T = P(.,.) with the roles swapped) instead of a trip through LDS whenever both operands are
// already in registers; what remains in LDS per step pair: two symmetrisations, two transposes, one row gather.
//
// ws record per (b,t), 544 floats: [ g mu_f (16) | g Sig_f (256) | g mu_p (16) | g Sig_p (256) ]; the two matrices are
// stored as their C-layout registers (store_rows / load_rows round-trip), they never leave this kernel.  The adjoints of
// Sig_f and Sig_s are only ever used through their symmetric part (G = sym(.), lgssm_bwd.h), so a summand may enter
// transposed.
// =====================================================================================================================
constexpr int WS_REC = 2 * (N + NN);
__device__ __forceinline__ f4 outer(f4 aW, float bL) { return f4{aW[0] * bL, aW[1] * bL, aW[2] * bL, aW[3] * bL}; }
__device__ __forceinline__ f4 zero4() { return f4{0.f, 0.f, 0.f, 0.f}; }

template <bool HAS_FP>   // HAS_FP: upstream gradients of the filtered / predicted stacks are present (all four)
__device__ __forceinline__ void rts_bwd_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                                              const kvae_lgssm_input_grads &G, float *ws, int b, Lds &L) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, T = P.T;
  const int64_t bT = (int64_t)b * T;
  float *w = ws + bT * WS_REC;
  // adjoint of the smoothed belief at t = 0 is the upstream gradient; the predicted belief at t = 0 only has upstream
  float gsmL = U.mus_smooth[bT * N + j];
  f4 gsS = load_rows(U.Sigmas_smooth + bT * NN, j, g);
  w[N + NN + j] = HAS_FP ? U.mus_pred[bT * N + j] : 0.0f;
  store_rows(w + N + NN + N, HAS_FP ? load_c(U.Sigmas_pred + bT * NN, j, g) : zero4(), j, g);
  store_rows(gstack_at(G.gA, b, 0), zero4(), j, g);
  struct In {
    f4 Sf, Spc, Spt, Ac, Ss, Jc, Jt, uSs, uSf, uSp;
    Row16 Spr;   // row j of Sig_p[t+1], whole: the matrix of the adjoint's solve, straight from memory instead of through LDS
    float mupL, musL, uMs, uMf, uMp;
  } s, nx;
  // running pointers of step t: Sig_f[t], (Sig_p, Sig_s, mu_p, mu_s, A, upstream)[t+1], J[t]
  int64_t q = bT;                 // index of (b, t)
  const float *pA = stack_at(P.A, b, T >= 2 ? 1 : 0);
  const int64_t sA = P.A.st;
  auto load = [&](In &o) {
    o.Sf = load_rows(S.Sigmas_filt + q * NN, j, g);
    o.Spc = load_c(S.Sigmas_pred + (q + 1) * NN, j, g);
    o.Spt = load_rows(S.Sigmas_pred + (q + 1) * NN, j, g);
    o.Spr = load_row16(S.Sigmas_pred + (q + 1) * NN, j);
    o.Ss = load_rows(S.Sigmas_smooth + (q + 1) * NN, j, g);
    o.Ac = load_c(pA, j, g);
    const float *J = S.aux + q * KV_AUX_N16 + N * P2 + P2 * P2;
    o.Jc = load_c(J, j, g);
    o.Jt = load_rows(J, j, g);
    o.mupL = S.mus_pred[(q + 1) * N + j];
    o.musL = S.mus_smooth[(q + 1) * N + j];
    o.uMs = U.mus_smooth[(q + 1) * N + j];
    o.uSs = load_rows(U.Sigmas_smooth + (q + 1) * NN, j, g);
    if constexpr (HAS_FP) {
      o.uMf = U.mus_filt[q * N + j];
      o.uSf = load_rows(U.Sigmas_filt + q * NN, j, g);
      o.uMp = U.mus_pred[(q + 1) * N + j];
      o.uSp = load_c(U.Sigmas_pred + (q + 1) * NN, j, g);
    }
  };
  if (T >= 2) load(s);
  nx = s;
  KV_DRAIN_VMEM();
  for (int t = 0; t + 1 < T; ++t) {
    if (t + 2 < T) q += 1, pA += sA;                                // unconditional prefetch (the last iteration re-reads its own step)
    load(nx);
    KV_PREFETCH_FENCE();
    float *wt = w + (int64_t)t * WS_REC;
    const f4 gM = symmetrise(gsS, L, j, g);
    const f4 D2 = (s.Ss - s.Spc) + (s.Ss - s.Spt);                  // D + D^T, D = Sig_s[t+1] - Sig_p[t+1]
    const f4 Y1 = mtn(gM, s.Jc);                                    // gM J
    const f4 Y1t = mtn(s.Jc, gM);                                   // J^T gM = Y1^T
    const f4 gD = mtn(s.Jc, Y1);                                    // J^T Y1
    const f4 gsmW = l2w(gsmL, lane);
    const float dmuL = s.musL - s.mupL;
    const f4 gJ = mtn(Y1t, D2, outer(gsmW, dmuL));                  // Y1 (D^T + D) + gsm dmu^T
    const float gdmL = mtv(s.Jc, gsmW);                             // J^T gsm
    float Sprow[N];
#pragma unroll
    for (int c = 0; c < N; ++c) Sprow[c] = s.Spr.q[c >> 2][c & 3];
    const f4 gRt = solve_rows(Sprow, gJ, lane);                     // Sig_p gR = gJ^T ; C-layout of gR^T
    const f4 gRc = transpose(gRt, L, j, g);
    const f4 gWA = mtn(gRc, s.Ac);                                  // gR^T A[t+1]
    const f4 gP = mtn(s.Jc, gRt);                                   // J^T gR^T
    const f4 gAs = mtn(gRt, s.Sf);                                  // gR Sig_f : the smoother's share of gA[t+1]
    store_rows(wt + N, (HAS_FP ? s.uSf : zero4()) + gM + gWA, j, g);
    store_rows(wt + WS_REC + N + NN + N, (HAS_FP ? s.uSp : zero4()) - gD - gP, j, g);
    store_rows(gstack_at(G.gA, b, t + 1), gAs, j, g);
    wt[j] = (HAS_FP ? s.uMf : 0.0f) + gsmL;
    wt[WS_REC + N + NN + j] = (HAS_FP ? s.uMp : 0.0f) - gdmL;
    gsS = s.uSs + gD;                                               // carried adjoint of the smoothed belief at t+1
    gsmL = s.uMs + gdmL;
    s = nx;
  }
  // t = T-1: smoothed == filtered
  float *wl = w + (int64_t)(T - 1) * WS_REC;
  const int64_t ql = bT + T - 1;
  store_rows(wl + N, (HAS_FP ? load_rows(U.Sigmas_filt + ql * NN, j, g) : zero4()) + gsS, j, g);
  wl[j] = (HAS_FP ? U.mus_filt[ql * N + j] : 0.0f) + gsmL;
}

template <bool HAS_GQ>
__device__ __forceinline__ void filter_bwd_sweep(const kvae_lgssm_problem &P, const kvae_lgssm_states &S,
                                                 const kvae_lgssm_input_grads &G, const float *ws, int b, Lds &L) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, T = P.T;
  const int64_t bT = (int64_t)b * T;
  const float R[4] = {P.R[0], P.R[1], P.R[2], P.R[3]};
  f4 Ic;
#pragma unroll
  for (int r = 0; r < 4; ++r) Ic[r] = j == 4 * g + r ? 1.0f : 0.0f;
  struct In {
    f4 At, Ac, Bc, Cw0, Cw1, Sigc, Sigt, Spc, Spt, kwa, kwb, Sv, wSf, wSp, gAs;
    float Cl0, Cl1, y0, y1, mk, uL, muL, mupL, kl0, kl1, wmf, wmp;
  } s, nx;
  StepPtr ptr;
  ptr.init(P, b);
  for (int t = 0; t + 1 < T; ++t) ptr.advance();                    // operands of step T-1
  int t_ld = T - 1;
  auto load = [&](In &o) {
    const int64_t q = bT + t_ld;
    o.At = load_rows(ptr.A, j, g);
    o.Ac = load_c(ptr.A, j, g);
    o.Bc = load_c(ptr.Bm, j, g);
    o.Cw0 = load_w(ptr.C, g), o.Cw1 = load_w(ptr.C + N, g);
    o.Cl0 = ptr.C[j], o.Cl1 = ptr.C[N + j];
    o.y0 = ptr.Y[0], o.y1 = ptr.Y[1];
    o.mk = *ptr.mk;
    o.uL = ptr.U[j];
    // belief at t-1 (the prior for t = 0); Sig_f is symmetric, the prior need not be: load both orientations
    const float *pS = t_ld > 0 ? S.Sigmas_filt + (q - 1) * NN : P.Sigma0 + (int64_t)b * P.Sigma0_sb;
    const float *pm = t_ld > 0 ? S.mus_filt + (q - 1) * N : P.mu0 + (int64_t)b * P.mu0_sb;
    o.Sigc = load_c(pS, j, g);
    o.Sigt = load_rows(pS, j, g);
    o.muL = pm[j];
    o.Spc = load_c(S.Sigmas_pred + q * NN, j, g);
    o.Spt = load_rows(S.Sigmas_pred + q * NN, j, g);
    o.mupL = S.mus_pred[q * N + j];
    const float *ax = S.aux + q * KV_AUX_N16;
    o.kwa = *reinterpret_cast<const f4 *>(ax + 8 * g), o.kwb = *reinterpret_cast<const f4 *>(ax + 8 * g + 4);
    o.kl0 = ax[2 * j], o.kl1 = ax[2 * j + 1];
    o.Sv = *reinterpret_cast<const f4 *>(ax + N * P2);
    const float *w = ws + q * WS_REC;
    o.wmf = w[j];
    o.wSf = load_rows(w + N, j, g);
    o.wmp = w[N + NN + j];
    o.wSp = load_rows(w + N + NN + N, j, g);
    o.gAs = load_rows(gstack_at(G.gA, b, t_ld), j, g);
  };
  load(s);
  nx = s;
  KV_DRAIN_VMEM();
  float gmuL = 0.0f;
  f4 gSig = zero4();
  for (int t = T - 1; t >= 0; --t) {
    if (t >= 1) {                                                    // unconditional prefetch of step t-1 (t = 0 re-reads itself)
      t_ld = t - 1;
      ptr.A -= ptr.sA, ptr.Bm -= ptr.sB, ptr.Q -= ptr.sQ, ptr.C -= ptr.sC, ptr.U -= N, ptr.Y -= P2, ptr.mk -= ptr.smk;
    }
    load(nx);
    KV_PREFETCH_FENCE();
    const int64_t q = bT + t;
    const float mk = P.mask ? s.mk : 1.0f;
    // total adjoint of the filtered belief at t = carried (from step t+1) + handed-off
    gmuL += s.wmf;
    gSig += s.wSf;
    const f4 Gm = symmetrise(gSig, L, j, g);
    // saved gains: K unmasked [16,2] (W: rows 4g..4g+3 interleaved; L: row j), S symmetric
    const f4 ku0 = f4{s.kwa[0], s.kwa[2], s.kwb[0], s.kwb[2]}, ku1 = f4{s.kwa[1], s.kwa[3], s.kwb[1], s.kwb[3]};
    const f4 kw0 = mk * ku0, kw1 = mk * ku1;
    const float kl0 = mk * s.kl0, kl1 = mk * s.kl1;
    f4 Mc;                                                           // C-layout of I - K C
#pragma unroll
    for (int r = 0; r < 4; ++r) Mc[r] = Ic[r] - (kw0[r] * s.Cl0 + kw1[r] * s.Cl1);
    const float gr0 = row_sum(kl0 * gmuL), gr1 = row_sum(kl1 * gmuL);   // gr = K^T gmu
    const float r0 = s.y0 - row_sum(s.Cl0 * s.mupL), r1 = s.y1 - row_sum(s.Cl1 * s.mupL);
    const f4 X1 = mtn(Gm, Mc);                                       // G (I - K C)
    const f4 X1t = mtn(Mc, Gm);                                      // its transpose
    const f4 Sp2 = s.Spc + s.Spt;                                    // Sig_p + Sig_p^T
    const f4 gIKC = mtn(X1t, Sp2);                                   // X1 (Sig_p^T + Sig_p)
    const f4 gIKCt = mtn(Sp2, X1t);                                  // its transpose
    f4 gSp = mtn(Mc, X1, s.wSp);                                     // (I - K C)^T X1 + handed-off adjoint of Sig_p
    // gK = G K (R^T + R) - gIKC C^T + gmu r^T, on lanes (row i = lane j)
    const float GK0 = mtv(Gm, kw0), GK1 = mtv(Gm, kw1);
    const float gK0 = GK0 * (R[0] + R[0]) + GK1 * (R[1] + R[2]) - mtv(gIKCt, s.Cw0) + gmuL * r0;
    const float gK1 = GK0 * (R[2] + R[1]) + GK1 * (R[3] + R[3]) - mtv(gIKCt, s.Cw1) + gmuL * r1;
    // gC (first part) = -K^T gIKC
    float gC0 = -mtv(gIKC, kw0), gC1 = -mtv(gIKC, kw1);
    // Z = S^{-T} (mask gK^T)   [2,16], column i on lane i
    const Inv2 F = factor2(s.Sv[0], s.Sv[1], s.Sv[3]);
    const Sol2 z = solve2f(F, mk * gK0, mk * gK1);
    // gS0 = sym(-Z Kt^T) with the UNMASKED gain; gCP = gS0 C
    const float zk00 = row_sum(z.x0 * s.kl0), zk01 = row_sum(z.x0 * s.kl1), zk10 = row_sum(z.x1 * s.kl0),
                zk11 = row_sum(z.x1 * s.kl1);
    const float h00 = -0.5f * (zk00 + zk00), h01 = -0.5f * (zk01 + zk10), h11 = -0.5f * (zk11 + zk11);
    const float gCPl0 = h00 * s.Cl0 + h01 * s.Cl1, gCPl1 = h01 * s.Cl0 + h11 * s.Cl1;
    const f4 gCPw0 = h00 * s.Cw0 + h01 * s.Cw1, gCPw1 = h01 * s.Cw0 + h11 * s.Cw1;
    const f4 zw0 = l2w(z.x0, lane), zw1 = l2w(z.x1, lane);
    // gSp += Z^T C + C^T gCP
    gSp += outer(zw0, s.Cl0) + outer(zw1, s.Cl1) + outer(s.Cw0, gCPl0) + outer(s.Cw1, gCPl1);
    // gC += Z Sig_p + gS0 (C Sig_p) + gCP Sig_p^T - gr mu_p^T
    const float cp0 = mtv(s.Spc, s.Cw0), cp1 = mtv(s.Spc, s.Cw1);
    gC0 += mtv(s.Spc, zw0) + (h00 * cp0 + h01 * cp1) + mtv(s.Spt, gCPw0) - gr0 * s.mupL;
    gC1 += mtv(s.Spc, zw1) + (h01 * cp0 + h11 * cp1) + mtv(s.Spt, gCPw1) - gr1 * s.mupL;
    float *gCo = gstack_at(G.gC, b, t);
    gCo[j] = gC0, gCo[N + j] = gC1;
    const float gmpL = gmuL + s.wmp - (s.Cl0 * gr0 + s.Cl1 * gr1);  // gmp = gmu + handed-off - C^T gr
    G.gY[q * P2 + (j & 1)] = (j & 1) ? gr1 : gr0;
    if constexpr (HAS_GQ) store_c(gstack_at(G.gQ, b, t), gSp, j, g);
    // gA[t] = smoother share + gSp^T (A Sig) + (gSp A) Sig^T + gmp mu^T ; carried adjoints for t-1
    const f4 gSpt = transpose(gSp, L, j, g);
    const f4 AS = mtn(s.At, s.Sigc);                                 // A Sig
    const f4 gASt = mtn(s.Ac, gSpt);                                 // A^T gSp^T = (gSp A)^T
    const f4 gAS = mtn(gSpt, s.Ac);                                  // gSp A
    const f4 gmpW = l2w(gmpL, lane);
    f4 gA = s.gAs + outer(gmpW, s.muL);
    gA = mtn(gSp, AS, gA);
    gA = mtn(gASt, s.Sigt, gA);
    store_c(gstack_at(G.gA, b, t), gA, j, g);
    gSig = mtn(s.Ac, gAS);                                           // A^T (gSp A)
    gmuL = mtv(s.Ac, gmpW);                                          // A^T gmp
    store_c(gstack_at(G.gB, b, t), outer(gmpW, s.uL), j, g);         // gB = gmp u^T
    G.gU[q * N + j] = mtv(s.Bc, gmpW);                               // gU = B^T gmp
    s = nx;
  }
  if (G.g_mu0) G.g_mu0[(int64_t)b * N + j] = gmuL;
  if (G.g_Sigma0) store_c(G.g_Sigma0 + (int64_t)b * NN, gSig, j, g);
}


// ---- the kernels' bodies (kvae_lgssm_n16.hip wraps them in __global__ functions with the tile in LDS; tests/hostsim runs them on
// emulated wavefronts): one sequence per wavefront, grid = B
template <bool AUX>
__device__ __forceinline__ void smooth_fwd_wave(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, int do_filter, int do_rts,
                                                Lds &L) {
  const int b = blockIdx.x;
  if (do_filter) {
    filter_sweep<AUX>(P, S, b, L);
    __syncthreads();   // the smoother reads back what this wavefront has just written
  }
  if (do_rts) rts_sweep<AUX>(P, S, b, L);
}
template <bool HAS_FP, bool HAS_GQ>
__device__ __forceinline__ void smooth_bwd_wave(const kvae_lgssm_problem &P, const kvae_lgssm_states &S, const kvae_lgssm_states &U,
                                                const kvae_lgssm_input_grads &G, float *ws, Lds &L) {
  const int b = blockIdx.x;
  rts_bwd_sweep<HAS_FP>(P, S, U, G, ws, b, L);
  __syncthreads();   // the filter sweep reads back the hand-off records this wavefront has just written
  filter_bwd_sweep<HAS_GQ>(P, S, G, ws, b, L);
}

}  // namespace n16
}  // namespace kvae
#endif
