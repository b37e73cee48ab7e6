// wave_emu.h — TEST-ONLY: run the wavefront-level kernel bodies (csrc/lgssm_q4.h, lgssm_m4.h, lgssm_n16.h: code written per lane
// around cross-lane instructions - DPP moves, permlane swaps, ds_bpermute, ballots, the f32 MFMA tiles) on the HOST, so that
// the CPU tier can put their indexing under AddressSanitizer (GPU sanitizers are not available on the pool).
//
// One emulated wavefront = 64 host threads, one per lane, running the very same body; every cross-lane intrinsic is a
// rendezvous: each lane deposits its operand, a barrier, each lane picks what the instruction would have handed it.  (Two
// operand buffers used alternately make one barrier per instruction enough: a lane can be at most one instruction ahead.)  The
// bodies emulated here only use such instructions under wave-uniform control flow.  (The n = 16 ELBO kernels do not: their four
// row-groups take turns on the matrix cores with DPP traffic inside group-dependent branches - legal on the hardware, where a
// DPP move only needs its own row active, a deadlock for this rendezvous; they are not emulated.)
// Arithmetic of the MFMA shims: k-ascending fmaf chains in fp32, which is what the kernels' parity argument assumes.
// This is a sanitizer / debug harness, slow by design (a barrier of 64 threads per instruction): tiny problems only.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <functional>
#include <thread>
#include <vector>

namespace wemu {

struct Wave {
  pthread_barrier_t bar;
  uint32_t a[2][64], b[2][64];
  Wave() { pthread_barrier_init(&bar, nullptr, 64); }
  ~Wave() { pthread_barrier_destroy(&bar); }
};
struct Dim3 { unsigned x, y, z; };
inline thread_local Wave *t_wave = nullptr;
inline thread_local int t_par = 0;
inline thread_local Dim3 t_tid = {0, 0, 0}, t_bid = {0, 0, 0}, t_gdim = {1, 1, 1};

inline void sync() { pthread_barrier_wait(&t_wave->bar); }
inline int lane_id() { return (int)(t_tid.x & 63); }
// every lane deposits v; returns the buffer to read the other lanes' values from (valid until this lane's next exchange)
inline const uint32_t *exchange(uint32_t v) {
  const int p = t_par;
  t_par ^= 1;
  t_wave->a[p][lane_id()] = v;
  sync();
  return t_wave->a[p];
}
inline uint32_t fbits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
inline float bitsf(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

// v_mov_b32_dpp with row_mask = bank_mask = 0xf, bound_ctrl: the controls the kernels use
inline int dpp_source(int lane, int ctrl) {
  if (ctrl >= 0 && ctrl <= 0xFF) return (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);   // quad_perm
  if (ctrl == 0x140) return (lane & ~15) | (15 - (lane & 15));                             // row_mirror
  if (ctrl == 0x141) return (lane & ~7) | (7 - (lane & 7));                                // row_half_mirror
  if (ctrl >= 0x150 && ctrl <= 0x15F) return (lane & ~15) | (ctrl - 0x150);                // row_newbcast:K
  __builtin_trap();   // a control this harness does not know: extend it rather than guess
}
inline int mov_dpp(int v, int ctrl, int, int, bool) {
  const uint32_t *all = exchange((uint32_t)v);
  return (int)all[dpp_source(lane_id(), ctrl)];
}
inline float shfl(float v, int src, int) { return bitsf(exchange(fbits(v))[src & 63]); }
inline unsigned long long ballot(bool pred) {
  const uint32_t *all = exchange(pred ? 1u : 0u);
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) m |= (unsigned long long)(all[l] & 1u) << l;
  return m;
}
inline bool any(bool pred) { return ballot(pred) != 0; }

struct U2 {
  uint32_t v[2];
  uint32_t operator[](int i) const { return v[i]; }
};
// v_permlane16_swap / v_permlane32_swap (fi = bc = false): {new vdst, new src0}; odd rows (halves) of vdst trade places with
// even rows (halves) of src0
inline U2 permlane_swap(uint32_t vdst, uint32_t src0, int width) {
  const int p = t_par;
  t_par ^= 1;
  const int l = lane_id();
  t_wave->a[p][l] = vdst, t_wave->b[p][l] = src0;
  sync();
  const bool upper = (l / width) & 1;
  U2 r;
  r.v[0] = upper ? t_wave->b[p][l - width] : vdst;    // vdst[upper] <- src0[lower]
  r.v[1] = upper ? src0 : t_wave->a[p][l + width];    // src0[lower] <- vdst[upper]
  return r;
}

template <class V4>
inline V4 mfma_4x4x1(float x, float y, V4 c) {          // 16 blocks of 4 lanes: D[r][j] += a(lane 4 blk + r) * b(lane 4 blk + j)
  const int p = t_par;
  t_par ^= 1;
  const int l = lane_id();
  t_wave->a[p][l] = fbits(x), t_wave->b[p][l] = fbits(y);
  sync();
  const int base = l & ~3;
  for (int r = 0; r < 4; ++r) c[r] = fmaf(bitsf(t_wave->a[p][base + r]), bitsf(t_wave->b[p][l]), c[r]);
  return c;
}
template <class V4>
inline V4 mfma_16x16x4(float x, float y, V4 c) {        // A[i][k] on lane i + 16 k, B[k][j] on lane j + 16 k, D[4 g + r][j] in reg r of lane (j, g)
  const int p = t_par;
  t_par ^= 1;
  const int l = lane_id(), j = l & 15, g = l >> 4;
  t_wave->a[p][l] = fbits(x), t_wave->b[p][l] = fbits(y);
  sync();
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 4; ++k) c[r] = fmaf(bitsf(t_wave->a[p][4 * g + r + 16 * k]), bitsf(t_wave->b[p][j + 16 * k]), c[r]);
  return c;
}

// launch: `blocks` one-wavefront workgroups, one after the other, 64 threads each
inline void launch(unsigned blocks, const std::function<void()> &body) {
  for (unsigned blk = 0; blk < blocks; ++blk) {
    Wave w;
    std::vector<std::thread> th;
    th.reserve(64);
    for (unsigned l = 0; l < 64; ++l)
      th.emplace_back([&, l] {
        t_wave = &w, t_par = 0, t_tid = {l, 0, 0}, t_bid = {blk, 0, 0}, t_gdim = {blocks, 1, 1};
        body();
      });
    for (auto &t : th) t.join();
  }
}

}  // namespace wemu

// ---- what the kernel headers see ----------------------------------------------------------------------------------------------
#define __device__
#define __forceinline__ inline
#define __global__
#define __launch_bounds__(n)
#define ext_vector_type(n) vector_size((n) * 4)   // g++: GNU vector types index, initialise and mix with scalars the same way
#define threadIdx (wemu::t_tid)
#define blockIdx (wemu::t_bid)
#define gridDim (wemu::t_gdim)
#define __syncthreads() wemu::sync()
#define __shfl(v, src, w) wemu::shfl((v), (src), (w))
#define __ballot(p) wemu::ballot((p))
#define __any(p) wemu::any((p))
#define __builtin_amdgcn_mov_dpp(v, ctrl, rm, bm, bc) wemu::mov_dpp((v), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, x, y, z) wemu::mfma_4x4x1((a), (b), (c))
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) wemu::mfma_16x16x4((a), (b), (c))
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) wemu::permlane_swap((a), (b), 16)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) wemu::permlane_swap((a), (b), 32)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sqrtf(x) sqrtf((x))
#define __logf(x) logf((x))
#define __expf(x) expf((x))
