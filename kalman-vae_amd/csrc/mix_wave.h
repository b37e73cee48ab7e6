// mix_wave.h — the mixture-of-K dynamics (mix.h: same equations; reference dyn_param.py:58-60, switch_dyn_param.py:82-84) as
// streaming kernels: ONE WAVEFRONT PER SLAB OF ROWS, the K base records in registers, every step record moved as 16-byte pieces.
//
// Why: at the configs[4] shard a step record is 3 KB (A | B | Q at n = 16: E = 768 floats) and the three mixing launches of round
// 2 moved it at 1.4-1.7 TB/s - 624 us per training step, as much as the smoother's forward - because the forward spent a 64-bit
// division per element, the backward read the gradient records TWICE (once for g_alpha, once for the base gradients) and wrote
// 59 MB of slab partials.  Here
//   forward   rec[r] = sum_k alpha[r,k] base[k]: the wavefront's lanes own the same 4-float pieces of every row; alpha[r,:] is a
//             wave-uniform load; nothing but FMAs and 16-byte stores in the loop;
//   backward  ONE pass over g_rec: g_alpha[r,k] = <g_rec[r], base[k]> (wave reduction) and the wavefront's share of
//             g_base[k] += alpha[r,k] g_rec[r] in registers, written once per wavefront (<= 1024 slab partials, summed by a
//             fixed-order second launch: run-to-run identical).
// E % 4 == 0, 128 <= E <= 768, K <= 8; anything else (n = 4: E = 40 / 48) takes the element-wise kernels of mix.h.
// configs[4] shard (102400 rows, E = 768, K = 3; profiles/r03_mix_probe_stats.txt): forward 190 -> 48 us (6.6 TB/s of stores),
// backward 224 + 102 + 108 -> 127 + 15 us.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {
namespace mixw {

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int MAX_SLABS = 2048;   // two wavefronts per SIMD; 2048 x K x E floats of partials at most

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// EJ 16-byte pieces per lane (piece j of lane l = floats 4 (64 j + l) .. + 3); grid = ceil(slabs / 4) blocks of 4 wavefronts
template <int KMAX, int EJ>
__global__ __launch_bounds__(256) void k_mix_fwd_wave(const float *__restrict__ alpha, const float *__restrict__ base,
                                                      float *__restrict__ out, int64_t rows, int K, int E, int64_t rows_per_slab) {
  const int lane = threadIdx.x & 63;
  const int64_t slab = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r0 = slab * rows_per_slab, r1 = r0 + rows_per_slab < rows ? r0 + rows_per_slab : rows;
  if (r0 >= rows) return;
  f4 b[KMAX][EJ];
  bool ok[EJ];
#pragma unroll
  for (int j = 0; j < EJ; ++j) {
    const int e = 4 * (64 * j + lane);
    ok[j] = e < E;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      b[k][j] = (k < K && ok[j]) ? *reinterpret_cast<const f4 *>(base + (int64_t)k * E + e) : (f4){0.f, 0.f, 0.f, 0.f};
  }
  for (int64_t r = r0; r < r1; ++r) {
    float a[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) a[k] = k < K ? alpha[r * K + k] : 0.f;
#pragma unroll
    for (int j = 0; j < EJ; ++j) {
      f4 o = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < KMAX; ++k)
        if (k < K) o = a[k] * b[k][j] + o;          // k ascending from zero: the element-wise kernel's order
      if (ok[j]) *reinterpret_cast<f4 *>(out + r * E + 4 * (64 * j + lane)) = o;
    }
  }
}

template <int KMAX, int EJ>
__global__ __launch_bounds__(256) void k_mix_bwd_wave(const float *__restrict__ alpha, const float *__restrict__ base,
                                                      const float *__restrict__ g_out, float *__restrict__ g_alpha,
                                                      float *__restrict__ partials, int64_t rows, int K, int E,
                                                      int64_t rows_per_slab, int nslab, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t slab = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slab >= nslab) return;
  const int64_t r0 = slab * rows_per_slab, r1 = r0 + rows_per_slab < rows ? r0 + rows_per_slab : rows;
  f4 b[KMAX][EJ], acc[KMAX][EJ];
  bool ok[EJ];
#pragma unroll
  for (int j = 0; j < EJ; ++j) {
    const int e = 4 * (64 * j + lane);
    ok[j] = e < E;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      b[k][j] = (k < K && ok[j]) ? *reinterpret_cast<const f4 *>(base + (int64_t)k * E + e) : (f4){0.f, 0.f, 0.f, 0.f};
      acc[k][j] = (f4){0.f, 0.f, 0.f, 0.f};
    }
  }
  auto load_row = [&](int64_t r, f4 (&g)[EJ]) {
#pragma unroll
    for (int j = 0; j < EJ; ++j) {
      const f4 v = *reinterpret_cast<const f4 *>(g_out + r * E + (ok[j] ? 4 * (64 * j + lane) : 0));   // always a valid address
      g[j] = ok[j] ? v : (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // RB rows per iteration: their K x RB partial dot products are reduced over the wavefront TOGETHER (independent butterfly
  // chains interleave; one row at a time left a wavefront waiting out six dependent cross-lane hops per dot product: 219 us at
  // the configs[4] shard with one wavefront per SIMD)
  constexpr int RB = 4;
  f4 g[RB][EJ];
  for (int64_t r = r0; r < r1; r += RB) {
#pragma unroll
    for (int u = 0; u < RB; ++u) load_row(r + u < r1 ? r + u : r1 - 1, g[u]);   // rows past the slab re-read its last row
    float d[RB][KMAX], a[RB][KMAX];
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        d[u][k] = 0.f;
        a[u][k] = (k < K && r + u < r1) ? alpha[(r + u) * K + k] : 0.f;
        if (k >= K) continue;
#pragma unroll
        for (int j = 0; j < EJ; ++j) {
          const f4 p = g[u][j] * b[k][j];
          d[u][k] += (p[0] + p[1]) + (p[2] + p[3]);
        }
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int u = 0; u < RB; ++u)
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
          if (k < K) d[u][k] += __shfl_xor(d[u][k], off, 64);
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k >= K) continue;
        if (lane == 0 && r + u < r1) g_alpha[(r + u) * K + k] = accumulate ? g_alpha[(r + u) * K + k] + d[u][k] : d[u][k];
#pragma unroll
        for (int j = 0; j < EJ; ++j) acc[k][j] = a[u][k] * g[u][j] + acc[k][j];   // (a = 0 for rows past the slab)
      }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int j = 0; j < EJ; ++j)
      if (k < K && ok[j]) *reinterpret_cast<f4 *>(partials + ((int64_t)slab * K + k) * E + 4 * (64 * j + lane)) = acc[k][j];
}

// g_base[i] = sum over the slab partials, 32 elements x 8 slab lanes per block: lane j sums slabs j, j + 8, ..., the eight sums
// are folded in lane order through LDS (a fixed order, whatever the launch)
__global__ __launch_bounds__(256) void k_mix_bwd_fold(const float *__restrict__ partials, float *__restrict__ g_base, int nslab,
                                                      int KE) {
  __shared__ float red[8][33];
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + el;
  float s = 0.f;
  if (i < KE)
    for (int k0 = sl; k0 < nslab; k0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = k0 + 8 * u < nslab ? partials[(int64_t)(k0 + 8 * u) * KE + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
  red[sl][el] = s;
  __syncthreads();
  if (sl != 0 || i >= KE) return;
  float t = red[0][el];
#pragma unroll
  for (int j = 1; j < 8; ++j) t += red[j][el];
  g_base[i] = t;
}

}  // namespace mixw
}  // namespace kvae
