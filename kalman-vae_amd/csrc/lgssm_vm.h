// lgssm_vm.h — execution model and small-matrix primitives shared by every LGSSM kernel body.
//
// Execution model ("one wavefront per problem"): a kernel body is written as a sequence of
// PHASES.  Inside a phase, KV_PAR(e, count) distributes `count` independent output elements over
// the 64 lanes of the wavefront; all operands live in LDS (per-wave scratch, struct *Lds below)
// so any lane may read any element written in an EARLIER phase.  KV_SYNC() ends a phase.
// Workgroups are exactly one wavefront (blockDim.x == 64), so KV_SYNC() is a wave-local LDS
// fence (s_waitcnt + s_barrier of a single wave), never a multi-wave rendezvous.
//
// Rules the bodies obey (so that lock-step lanes and the host simulation agree):
//   * an element written in a phase is never read in the same phase by another element;
//   * no in-place update reads neighbours of the array it writes;
//   * code outside KV_PAR is executed redundantly by every lane and must be side-effect free
//     except under KV_LANE0.
//
// KVAE_HOSTSIM builds the same bodies for the host (tests/hostsim): KV_PAR becomes a serial
// loop, so the arithmetic of every kernel can be run under ASan/UBSan and compared with the
// oracle in the CPU-only test tier.  It is a sanitizer/debug harness, not a product path: the
// Python package only ever loads the gfx950 library.
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/kvae_lgssm.h"

#if defined(KVAE_HOSTSIM)
#define KV_DEV static inline
#define KV_MEM inline
#define KV_LANES 1
#define KV_LANE 0
#define KV_SYNC() ((void)0)
#define KV_UNROLL
#elif defined(KV_TPP)
// kvae_lgssm_tpp.hip: ONE THREAD per problem on the GPU — the same bodies as the host simulation (KV_PAR is a serial
// loop, the *Lds struct is thread-private), 64 independent problems per wavefront.  For bodies without recursion in t.
#include <hip/hip_runtime.h>
#define KV_DEV __device__ __forceinline__
#define KV_MEM __device__ __forceinline__
#define KV_LANES 1
#define KV_LANE 0
#define KV_SYNC() ((void)0)
#define KV_UNROLL _Pragma("unroll")
#else
#include <hip/hip_runtime.h>
#define KV_DEV __device__ __forceinline__
#define KV_MEM __device__ __forceinline__
#ifndef KV_LANES
#define KV_LANES 64   // threads per problem: 64 = one wavefront; kvae_lgssm_wide.hip builds the bodies with 256
#endif
#define KV_LANE ((int)threadIdx.x)
#define KV_SYNC() __syncthreads()
#define KV_UNROLL _Pragma("unroll")
#endif
#if defined(KV_TPP)   // serial loops with compile-time bounds: unroll so that the thread-private state becomes registers
#define KV_PAR(e, count) _Pragma("unroll") for (int e = 0; e < (count); ++e)
#else
#define KV_PAR(e, count) for (int e = KV_LANE; e < (count); e += KV_LANES)
#endif
#define KV_LANE0 if (KV_LANE == 0)
#define KV_PF_SLOTS(CNT) (((CNT) + KV_LANES - 1) / KV_LANES)

namespace kvae {

// ---- problem dimensions: compile-time (specialised kernels) or run-time (generic kernel) ----
template <int N_, int M_, int P_>
struct SDims {
  static constexpr bool is_static = true;
  static constexpr int NMAX = N_, MMAX = M_, PMAX = P_;
  KV_MEM SDims(int, int, int) {}
  KV_MEM constexpr int n() const { return N_; }
  KV_MEM constexpr int m() const { return M_; }
  KV_MEM constexpr int p() const { return P_; }
};

struct RDims {
  static constexpr bool is_static = false;
  static constexpr int NMAX = KVAE_MAX_DIM, MMAX = KVAE_MAX_DIM, PMAX = KVAE_MAX_DIM;
  int n_, m_, p_;
  KV_MEM RDims(int n, int m, int p) : n_(n), m_(m), p_(p) {}
  KV_MEM int n() const { return n_; }
  KV_MEM int m() const { return m_; }
  KV_MEM int p() const { return p_; }
};

// ---- register prefetch of one per-step operand (global -> regs now, regs -> LDS later) -------
template <int CNT>
struct Prefetch {
  float v[KV_PF_SLOTS(CNT)];
  KV_MEM void issue(const float *g, int cnt) {
    KV_UNROLL
    for (int s = 0; s < KV_PF_SLOTS(CNT); ++s) {
      const int e = s * KV_LANES + KV_LANE;
      if (e < cnt) v[s] = g[e];
    }
  }
  KV_MEM void commit(float *lds, int cnt) const {
    KV_UNROLL
    for (int s = 0; s < KV_PF_SLOTS(CNT); ++s) {
      const int e = s * KV_LANES + KV_LANE;
      if (e < cnt) lds[e] = v[s];
    }
  }
};

KV_DEV void copy_in(float *lds, const float *g, int cnt) {
  KV_PAR(e, cnt) { lds[e] = g[e]; }
}
KV_DEV void copy_out(float *g, const float *lds, int cnt) {
  KV_PAR(e, cnt) { g[e] = lds[e]; }
}

// ---- packed row permutation (<= 16 rows, 4 bits each) kept redundantly in every lane ---------
struct Perm {
  uint64_t bits;
  KV_MEM Perm() : bits(0xFEDCBA9876543210ull) {}
  KV_MEM int get(int i) const { return (int)((bits >> (4 * i)) & 15ull); }
  KV_MEM void swap(int i, int j) {
    const uint64_t a = (bits >> (4 * i)) & 15ull, b = (bits >> (4 * j)) & 15ull;
    bits &= ~((15ull << (4 * i)) | (15ull << (4 * j)));
    bits |= (b << (4 * i)) | (a << (4 * j));
  }
};

// LU with partial pivoting on the augmented matrix aug[r x (r+nrhs)] (row-major, ld = r+nrhs),
// followed by back substitution; the solution X (r x nrhs) is written to out[i*ldo + j].
// Mirrors what torch.linalg.solve does (getrf + getrs): first-maximum pivot, row exchanges.
// One phase per pivot column + one for the substitution.  Ends with KV_SYNC().
KV_DEV void lu_solve(float *aug, int r, int nrhs, float *out, int ldo) {
  const int ld = r + nrhs;
  Perm pm;
  for (int c = 0; c < r; ++c) {
    int piv = c;
    float best = fabsf(aug[pm.get(c) * ld + c]);
    for (int i = c + 1; i < r; ++i) {
      const float v = fabsf(aug[pm.get(i) * ld + c]);
      if (v > best) { best = v; piv = i; }
    }
    if (piv != c) pm.swap(c, piv);
    const int pc = pm.get(c);
    const float rinv = 1.0f / aug[pc * ld + c];
    const int rows = r - c - 1, cols = ld - c - 1;
    KV_PAR(e, rows * cols) {
      const int ii = e / cols, jj = e - ii * cols;
      const int pi = pm.get(c + 1 + ii), j = c + 1 + jj;
      const float l = aug[pi * ld + c] * rinv;
      aug[pi * ld + j] = fmaf(-l, aug[pc * ld + j], aug[pi * ld + j]);
    }
    KV_SYNC();
  }
  KV_PAR(j, nrhs) {
    // U x = y, rows taken through the permutation; serial per right-hand side
    for (int c = r - 1; c >= 0; --c) {
      const int pc = pm.get(c);
      float acc = aug[pc * ld + r + j];
      for (int k = c + 1; k < r; ++k) acc = fmaf(-aug[pc * ld + k], out[k * ldo + j], acc);
      out[c * ldo + j] = acc / aug[pc * ld + c];
    }
  }
  KV_SYNC();
}

// Lower Cholesky of a[n x n] (symmetric, row-major) into Lo (upper part zeroed), left-looking
// like LAPACK potf2.  Returns false (uniformly in every lane) when a pivot is <= 0 or NaN.
// One phase per column.  Ends with KV_SYNC().
KV_DEV bool cholesky(const float *a, float *Lo, int n) {
  KV_PAR(e, n * n) {
    const int i = e / n, j = e - i * n;
    if (j > i) Lo[e] = 0.0f;
  }
  bool ok = true;
  for (int c = 0; c < n; ++c) {
    float d = a[c * n + c];
    for (int k = 0; k < c; ++k) d = fmaf(-Lo[c * n + k], Lo[c * n + k], d);
    if (!(d > 0.0f)) { ok = false; break; }
    const float sd = sqrtf(d);
    KV_PAR(i, n - c) {
      const int row = c + i;
      if (i == 0) {
        Lo[c * n + c] = sd;
      } else {
        float s = a[row * n + c];
        for (int k = 0; k < c; ++k) s = fmaf(-Lo[row * n + k], Lo[c * n + k], s);
        Lo[row * n + c] = s / sd;
      }
    }
    KV_SYNC();
  }
  KV_SYNC();
  return ok;
}

// x <- L^{-1} x for ncols right-hand sides stored as x[i*ldx + j]; serial per column.
KV_DEV void trisolve_lower(const float *Lm, int n, float *x, int ldx, int ncols) {
  KV_PAR(j, ncols) {
    for (int i = 0; i < n; ++i) {
      float acc = x[i * ldx + j];
      for (int k = 0; k < i; ++k) acc = fmaf(-Lm[i * n + k], x[k * ldx + j], acc);
      x[i * ldx + j] = acc / Lm[i * n + i];
    }
  }
}
// x <- L^{-T} x
KV_DEV void trisolve_lower_t(const float *Lm, int n, float *x, int ldx, int ncols) {
  KV_PAR(j, ncols) {
    for (int i = n - 1; i >= 0; --i) {
      float acc = x[i * ldx + j];
      for (int k = i + 1; k < n; ++k) acc = fmaf(-Lm[k * n + i], x[k * ldx + j], acc);
      x[i * ldx + j] = acc / Lm[i * n + i];
    }
  }
}

// The jitter ladder of KalmanFilter._safe_cholesky (kalman_filter.py:289-296): 1e-6, then *10
// per retry in Python double arithmetic, rounded to fp32 when multiplied into eye().
KV_DEV float jitter_of_level(int level) {
  double j = 1e-6;
  for (int i = 0; i < level; ++i) j *= 10.0;
  return (float)j;
}

}  // namespace kvae
