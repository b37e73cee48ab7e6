// vae_conv_mid.h — the encoder's two stride-2 layers, Conv2d(32 -> 32, 3x3, stride 2, pad 1) + ReLU on 16x16 and on
// 8x8 frames (reference kvae/vae/vae.py:20-31), as implicit GEMMs on the exact-f32 matrix cores
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, so the numerics stay plain fp32).
//
// Why hand-written: stride 2 rules out Winograd, and the library's fallbacks run these layers at ~36 TFLOP/s
// (forward 359 us, data gradient 470 us, weight gradient 140 us + NHWC transposes at 12800 frames) where the f32 MFMA
// peak (157 TFLOP/s) and the HBM traffic both put them near 100 us.  All three passes share one shape:
//   * one persistent workgroup (4 waves, one per SIMD) per CU; an "iteration" is 64 KiB of input frames
//     (2 frames of 16x16 or 8 frames of 8x8) = 128 output pixels = 4 MFMA column tiles of 32, one per wave;
//   * the 36 KiB of weights sit in LDS for the whole kernel, laid out [tap][k][m] so that both operands of every
//     MFMA are one ds_read_b32 with a compile-time offset; 144 MFMAs per wave per iteration;
//   * the next iteration's frames are fetched into registers while the current one is on the matrix cores;
//   * bias + ReLU (forward) and the ReLU mask (both gradients) are fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {

typedef float em_f16 __attribute__((ext_vector_type(16)));
constexpr int EM_C = 32, EM_W = EM_C * EM_C * 9;   // channels (in == out), weights

template <int S>
struct EmDims {
  static constexpr int OS = S / 2, PF = OS * OS;             // output side, output pixels per frame
  static constexpr int FPI = 128 / PF;                       // frames per iteration (2 or 8)
  static constexpr int PLANE = S * S, FRAME = EM_C * PLANE;  // input floats per channel plane / per frame
  static constexpr int IT_IN = FPI * FRAME;                  // 16384 floats in per iteration
  static constexpr int IT_OUT = FPI * EM_C * PF;             // 4096 floats out per iteration
};

// Wl[(tap*32 + k)*32 + m]: forward m = co, k = ci; data gradient m = ci, k = co.
__device__ __forceinline__ void em_load_weights(float *Wl, const float *__restrict__ W, bool m_is_co) {
  for (int i = threadIdx.x; i < EM_W; i += 256) {
    const int tap = i >> 10, k = (i >> 5) & 31, m = i & 31;
    const int co = m_is_co ? m : k, ci = m_is_co ? k : m;
    Wl[i] = W[(co * EM_C + ci) * 9 + tap];
  }
}

constexpr int EM_OOB = 1 << 24;   // float index 64 MiB past the start of LDS
constexpr int64_t EM_MAX_BYTES = (int64_t)1 << 31;   // one launch addresses its tensors with 32-bit byte offsets

// Global traffic goes through raw buffer instructions: out-of-range lanes (the ragged last iteration, the prefetch
// past the end) load zeros / drop their stores in hardware, so the loops carry no branches around memory operations
// and hipcc's wait-count insertion can count loads and stores exactly instead of draining vmcnt to 0.
typedef unsigned int em_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t em_rsrc(const void *p, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 em_ld4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  const em_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void em_st4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, const float4 v) {
  const em_u4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, r, byte_off, 0, 0);
}
__device__ __forceinline__ void em_st1(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, byte_off, 0, 0);
}
#ifdef KVAE_EM_STAMPS   // tools/em_stamp.hip only: per-phase s_memtime stamps of workgroup 0, thread 0
__device__ unsigned long long em_stamps[4096];
#define EM_STAMP(slot, k) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (slot) < 500) em_stamps[(slot) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define EM_STAMP(slot, k) do {} while (0)
#endif
#define KV_MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// row of accumulator register r on this lane half (C/D map of the 32x32 shapes)
#define KV_ACC_ROW(r, half) (((r) & 3) + 8 * ((r) >> 2) + 4 * (half))

// ---------------------------------------------------------------------------------------------------------------
// forward: out[n,co,oh,ow] = relu(b[co] + sum_{ci,ky,kx} W[co,ci,ky,kx] in[n,ci,2oh+ky-1,2ow+kx-1])
// GEMM view per wave: D[co][pixel] (32 x 32) += A[co][k] B[k][pixel], k = (tap, ci), 288 deep.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_enc_mid_fwd(const float *__restrict__ in, const float *__restrict__ W,
                                                     const float *__restrict__ bias, float *__restrict__ out, int64_t N) {
  using D = EmDims<S>;
  __shared__ float lds[EM_W + D::IT_IN + D::IT_OUT];
  float *Wl = lds, *fr = lds + EM_W, *ot = fr + D::IT_IN;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane & 31, half = lane >> 5;
  em_load_weights(Wl, W, true);
  const int64_t iters = (N + D::FPI - 1) / D::FPI, total = N * D::FRAME;
  const int gp = 32 * wv + q, fl = gp / D::PF, pix = gp % D::PF, oh = pix / D::OS, ow = pix % D::OS;
  const bool top = oh == 0, left = ow == 0;
  const int bbase = fl * D::FRAME + half * D::PLANE + (2 * oh - 1) * S + (2 * ow - 1);
  // Row -1 / column -1 of the padding: those lanes read from far outside the LDS allocation, where DS reads return 0
  // (a per-MFMA select instead costs 12 % of the kernel).  One base per border case, chosen once.
  const int b_t = top ? EM_OOB : bbase, b_l = left ? EM_OOB : bbase, b_tl = (top || left) ? EM_OOB : bbase;
  const int abase = half * 32 + q;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias[KV_ACC_ROW(r, half)];

  // Workgroups walk their 64 KiB in 4 KiB pieces starting at different pieces, so that at any instant the CUs are
  // spread over the memory channels instead of all sitting on the same offset of their chunk.
  const int rot = blockIdx.x & 15;
  const __amdgpu_buffer_rsrc_t rin = em_rsrc(in, total * 4), rout = em_rsrc(out, N * EM_C * D::PF * 4);
  float4 pre[16];
  int64_t it = blockIdx.x;
  auto fetch1 = [&](int64_t i, int j) {
    pre[j] = em_ld4(rin, (uint32_t)(i * D::IT_IN + (tid + 256 * ((j + rot) & 15)) * 4) * 4u);
  };
  // An iteration's 16 KiB of results go through an LDS image (same layout as the global chunk) and leave as four
  // dwordx4 stores per thread at the START of the next iteration, before its prefetch: 16 scattered dword stores per
  // lane are store-issue-bound (~1.3 us per iteration), and stores issued right before the loop-top wait would put
  // their latency on the critical path.
  // Vector-memory instructions are NOT fire-and-forget for the issuing wave: 16 dwordx4 loads issued back to back
  // hold it for ~3000 cycles (measured with s_memtime; the CU's memory pipeline takes them at ~21 B/clk), and with one
  // wave per SIMD the matrix core idles meanwhile.  So the 4 stores and 16 loads of an iteration are dealt out one per
  // MFMA group (every 256 cycles) instead.
  uint32_t done_base = 0x80000000u;                     // out of range: nothing to store yet
  auto flush1 = [&](int j) {
    em_st4(rout, done_base + (uint32_t)(tid + 256 * j) * 16u, reinterpret_cast<const float4 *>(ot)[tid + 256 * j]);
  };
#pragma unroll
  for (int j = 0; j < 16; ++j) fetch1(it, j);
  for (int slot = 0; it < iters; it += gridDim.x, ++slot) {
    EM_STAMP(slot, 0);
    __syncthreads();
    EM_STAMP(slot, 1);
#pragma unroll
    for (int j = 0; j < 16; ++j) reinterpret_cast<float4 *>(fr)[tid + 256 * ((j + rot) & 15)] = pre[j];
    EM_STAMP(slot, 2);
    __syncthreads();
    EM_STAMP(slot, 3);
    EM_STAMP(slot, 4);
    em_f16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // 144 MFMAs as 36 groups of 4; the operands of group g+2 are read while group g is on the matrix core
    // (sched_barrier keeps hipcc from sinking the reads back next to their use, which exposes the LDS latency).
    float av[36][4], bw[36][4];
    auto rd = [&](int g) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = g * 4 + u, tap = k / 16, jj = k % 16, ky = tap / 3, kx = tap % 3;
        av[g][u] = Wl[abase + (tap * 32 + 2 * jj) * 32];
        bw[g][u] = fr[(ky == 0 ? (kx == 0 ? b_tl : b_t) : (kx == 0 ? b_l : bbase)) + ky * S + kx + 2 * jj * D::PLANE];
      }
    };
    rd(0);
    rd(1);
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      if (g + 2 < 36) rd(g + 2);
      if (g < 4) flush1(g);                             // previous image out first (oldest in the vmcnt order)
      else if (g < 20) fetch1(it + gridDim.x, g - 4);   // past the end: zeros, no branch
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = KV_MFMA_F32(av[g][u], bw[g][u], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    EM_STAMP(slot, 5);
    __syncthreads();                                    // every wave has read the previous image out (flush above)
    EM_STAMP(slot, 6);
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[(fl * EM_C + KV_ACC_ROW(r, half)) * D::PF + pix] = fmaxf(acc[r] + bv[r], 0.f);
    EM_STAMP(slot, 7);
    done_base = (uint32_t)(it * D::IT_OUT) * 4u;        // frames >= N: dropped by the hardware
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) flush1(j);
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient: g_in[n,ci,ih,iw] = sum_{co,ky,kx : ih = 2oh+ky-1, iw = 2ow+kx-1} W[co,ci,ky,kx] gm[n,co,oh,ow],
// gm = g_out * (out > 0).  With stride 2 an input pixel sees only the taps of its row/column parity: the four parity
// classes (ph,pw) are four GEMMs of depth 32 x {1,2,2,4} taps over the 128 class pixels of an iteration; a wave runs
// all four for its 32 class pixels (144 MFMAs), and the result goes out through an LDS image of the frames so that
// the global stores are whole rows.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_enc_mid_bwd_data(const float *__restrict__ W, const float *__restrict__ out,
                                                          const float *__restrict__ g_out, float *__restrict__ g_in, int64_t N) {
  using D = EmDims<S>;
  __shared__ float lds[EM_W + D::IT_OUT + D::IT_IN];
  float *Wl = lds, *gm = lds + EM_W, *ot = gm + D::IT_OUT;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane & 31, half = lane >> 5;
  em_load_weights(Wl, W, false);
  const int64_t iters = (N + D::FPI - 1) / D::FPI, total_in = N * D::FRAME, total_out = N * EM_C * D::PF;
  const int cp = 32 * wv + q, fl = cp / D::PF, idx = cp % D::PF, a_ = idx / D::OS, b_ = idx % D::OS;
  const bool bottom = a_ == D::OS - 1, right = b_ == D::OS - 1;
  const int gbase = fl * EM_C * D::PF + half * D::PF + a_ * D::OS + b_;
  const int g_b = bottom ? EM_OOB : gbase, g_r = right ? EM_OOB : gbase, g_br = (bottom || right) ? EM_OOB : gbase;
  const int abase = half * 32 + q;
  const int obase = fl * D::FRAME + (2 * a_) * S + 2 * b_;

  const int rot = blockIdx.x & 15;
  const __amdgpu_buffer_rsrc_t rg = em_rsrc(g_out, total_out * 4), ro = em_rsrc(out, total_out * 4),
                               rgi = em_rsrc(g_in, total_in * 4);
  float4 pg[4], po[4];
  int64_t it = blockIdx.x;
  auto fetch1 = [&](int64_t i, int j) {                 // j < 8: g_out pieces 0..3, then out pieces 0..3
    const uint32_t e = (uint32_t)(i * D::IT_OUT + (tid + 256 * (((j & 3) + rot) & 3)) * 4) * 4u;
    if (j < 4) pg[j] = em_ld4(rg, e);
    else po[j - 4] = em_ld4(ro, e);
  };
  // The image of iteration i leaves during iteration i+1: its 16 stores and the 8 prefetch loads are dealt out one per
  // MFMA group (see the forward kernel).  The deepest class (4 taps, 16 groups) runs first so that all 16 stores have
  // read the old image before the first class result is written into it.
  uint32_t done_base = 0x80000000u;                     // out of range until an image exists
  auto flush1 = [&](int j) {
    const int c = (j + rot) & 15;
    em_st4(rgi, done_base + (uint32_t)(tid + 256 * c) * 16u, reinterpret_cast<const float4 *>(ot)[tid + 256 * c]);
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) fetch1(it, j);
  for (; it < iters; it += gridDim.x) {
    __syncthreads();                                   // previous iteration's gm consumed, its ot complete
#pragma unroll
    for (int j = 0; j < 4; ++j)
      reinterpret_cast<float4 *>(gm)[tid + 256 * ((j + rot) & 3)] =
          make_float4(po[j].x > 0.f ? pg[j].x : 0.f, po[j].y > 0.f ? pg[j].y : 0.f, po[j].z > 0.f ? pg[j].z : 0.f,
                      po[j].w > 0.f ? pg[j].w : 0.f);
    __syncthreads();                                   // gm visible
    // nine (class, tap) entries, deepest class first, 16 MFMAs each, run as 36 groups of 4 with the operands of group
    // g+2 in flight while group g is on the matrix core
    constexpr int E_CLS[9] = {3, 3, 3, 3, 1, 1, 2, 2, 0};
    constexpr int E_KY[9] = {0, 0, 2, 2, 1, 1, 0, 2, 1};
    constexpr int E_KX[9] = {0, 2, 0, 2, 0, 2, 1, 1, 1};
    float av[36][4], bw[36][4];
    auto rd = [&](int g) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int en = g / 4, jj = (g % 4) * 4 + u, ph = E_CLS[en] >> 1, pw = E_CLS[en] & 1;
        const int dy = (ph == 1 && E_KY[en] == 0) ? 1 : 0, dx = (pw == 1 && E_KX[en] == 0) ? 1 : 0;   // oh = a + dy
        av[g][u] = Wl[abase + ((E_KY[en] * 3 + E_KX[en]) * 32 + 2 * jj) * 32];
        bw[g][u] = gm[(dy ? (dx ? g_br : g_b) : (dx ? g_r : gbase)) + 2 * jj * D::PF + dy * D::OS + dx];
      }
    };
    rd(0);
    rd(1);
    em_f16 acc;
#pragma unroll
    for (int g = 0; g < 36; ++g) {
      const int en = g / 4, cls = E_CLS[en], ph = cls >> 1, pw = cls & 1;
      if (g % 4 == 0 && (en == 0 || E_CLS[en - 1] != cls)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
      if (g + 2 < 36) rd(g + 2);
      if (g < 16) flush1(g);                            // old image out (oldest in the vmcnt order)
      else if (g < 24) fetch1(it + gridDim.x, g - 16);  // past the end: zeros, no branch
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = KV_MFMA_F32(av[g][u], bw[g][u], acc);
      __builtin_amdgcn_sched_barrier(0);
      if (g % 4 == 3 && (en == 8 || E_CLS[en + 1] != cls)) {   // class complete: rows ci, column = this lane's pixel
        if (g == 15) __syncthreads();                   // every wave has read the old image (16 flushes) before it changes
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[obase + KV_ACC_ROW(r, half) * D::PLANE + ph * S + pw] = acc[r];
      }
    }
    done_base = (uint32_t)(it * D::IT_IN) * 4u;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) flush1(j);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co,ci,tap] = sum_{n,oh,ow} gm[n,co,oh,ow] in[n,ci,2oh+ky-1,2ow+kx-1]; db[co] = sum gm.
// GEMM view: nine 32 x 32 tiles D_tap[co][ci] += A[co][pixel] B_tap[pixel][ci], 819k pixels deep.  The 64 pixel pairs
// of an iteration are split over the four waves (16 each, x 9 taps = 144 MFMAs); every wave keeps its own nine tiles
// in 144 accumulator registers for the whole kernel; at the end the workgroup folds them into one partial row.
// Channel planes are padded by one float in LDS so that 32 lanes on 32 channels hit 32 banks.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_enc_mid_wrw(const float *__restrict__ in, const float *__restrict__ out,
                                                     const float *__restrict__ g_out, float *__restrict__ w_partials,
                                                     float *__restrict__ b_partials, int64_t N) {
  using D = EmDims<S>;
  constexpr int CSP = D::PLANE + 1, GSP = D::PF + 1;
  __shared__ float lds[D::FPI * EM_C * GSP + D::FPI * EM_C * CSP];
  float *gm = lds, *xin = lds + D::FPI * EM_C * GSP;   // gm first: the (discarded) row/column -1 reads of xin stay in LDS
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane & 31, half = lane >> 5;
  const int64_t iters = (N + D::FPI - 1) / D::FPI, total_in = N * D::FRAME, total_out = N * EM_C * D::PF;
  // this wave's 16 pixel pairs: S = 16 -> frame wv/2, pixels [32*(wv&1), +32); S = 8 -> frames 2wv, 2wv+1, all 16 pixels
  const int wframe = S == 16 ? (wv >> 1) : 2 * wv, wpix0 = S == 16 ? 32 * (wv & 1) : 0;
  const int oh0 = wpix0 / D::OS;
  const int abase = wframe * EM_C * GSP + q * GSP + wpix0 + half;
  const int bbase = wframe * EM_C * CSP + q * CSP + (2 * oh0 - 1) * S - 1 + 2 * half;
  // row -1 (first pixel row of the frame, wave-uniform) and column -1 (first pixel of a pair at column 0): read zeros
  // from outside the LDS allocation instead of selecting per MFMA
  const bool wtop = oh0 == 0, wleft = half == 0;
  const int x_t = wtop ? EM_OOB : bbase, x_l = wleft ? EM_OOB : bbase, x_tl = (wtop || wleft) ? EM_OOB : bbase;

  em_f16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int rot = blockIdx.x & 15;
  const __amdgpu_buffer_rsrc_t rin = em_rsrc(in, total_in * 4), rg = em_rsrc(g_out, total_out * 4),
                               ro = em_rsrc(out, total_out * 4);
  float4 px[16], pg[4], po[4];
  int64_t it = blockIdx.x;
  // Which 16 bytes of an input piece a thread moves.  With 256-float channel planes (S = 16) consecutive lanes would stay inside
  // ONE plane, and the four scalar LDS writes per 16 bytes (planes are padded by one float for the B-operand reads) then hit 8
  // banks with 64 lanes: an 8-way conflict on 64 writes per thread and iteration.  Sixteen lanes per plane, four planes per wave
  // instead: 32 banks, two lanes each; the loads stay 256-byte contiguous per 16 lanes.
  const int tin = S == 16 ? ((lane >> 4) << 6) | (wv << 4) | (lane & 15) : tid;
  auto fetch1 = [&](int64_t i, int j) {                 // 24 pieces: 16 of in, 4 of g_out, 4 of out
    if (j < 16) {
      px[j] = em_ld4(rin, (uint32_t)(i * D::IT_IN + (tin + 256 * ((j + rot) & 15)) * 4) * 4u);
    } else {
      const uint32_t e = (uint32_t)(i * D::IT_OUT + (tid + 256 * (((j & 3) + rot) & 3)) * 4) * 4u;
      if (j < 20) pg[j - 16] = em_ld4(rg, e);
      else po[j - 20] = em_ld4(ro, e);
    }
  };
#pragma unroll
  for (int j = 0; j < 24; ++j) fetch1(it, j);
  for (; it < iters; it += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = (tin + 256 * ((j + rot) & 15)) * 4, f = e / D::FRAME, c = (e % D::FRAME) / D::PLANE, p = e % D::PLANE;
      float *d = xin + (f * EM_C + c) * CSP + p;
      d[0] = px[j].x; d[1] = px[j].y; d[2] = px[j].z; d[3] = px[j].w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = (tid + 256 * ((j + rot) & 3)) * 4, f = e / (EM_C * D::PF), c = (e / D::PF) % EM_C, p = e % D::PF;
      float *d = gm + (f * EM_C + c) * GSP + p;
      d[0] = po[j].x > 0.f ? pg[j].x : 0.f; d[1] = po[j].y > 0.f ? pg[j].y : 0.f;
      d[2] = po[j].z > 0.f ? pg[j].z : 0.f; d[3] = po[j].w > 0.f ? pg[j].w : 0.f;
    }
    __syncthreads();
    // pixel pair i of this wave: frame offset fi, first pixel p (even) relative to wpix0; the ten operands of pair
    // i+1 are read while the nine MFMAs of pair i run
    float av[16], bw[16][9];
    auto rd = [&](int i) {
      const int fi = S == 16 ? 0 : i / 8, p = S == 16 ? 2 * i : 2 * (i % 8);
      const int ohr = p / D::OS, owr = p % D::OS;          // relative row, column of the pair's first pixel
      av[i] = gm[abase + fi * EM_C * GSP + p];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bool zt = tap / 3 == 0 && ohr == 0, zl = tap % 3 == 0 && owr == 0;
        bw[i][tap] = xin[(zt ? (zl ? x_tl : x_t) : (zl ? x_l : bbase)) + fi * EM_C * CSP + (2 * ohr + tap / 3) * S + 2 * owr + tap % 3];
      }
    };
    rd(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i + 1 < 16) rd(i + 1);
      if (i < 12) {                                       // the next iteration's 24 loads, two per group (see forward)
        fetch1(it + gridDim.x, 2 * i);
        fetch1(it + gridDim.x, 2 * i + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      bsum += av[i];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) acc[tap] = KV_MFMA_F32(av[i], bw[i][tap], acc[tap]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // D_tap[co][ci]: column ci = q on the lane, rows co in the registers.  The four waves fold their tiles through LDS
  // (one after the other) so that a workgroup emits ONE partial row, written coalesced.
  float *red = lds, *redb = lds + EM_W;
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wv == w) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float *d = red + (KV_ACC_ROW(r, half) * EM_C + q) * 9 + tap;
          *d = w == 0 ? acc[tap][r] : *d + acc[tap][r];
        }
      float *db = redb + half * EM_C + q;
      *db = w == 0 ? bsum : *db + bsum;
    }
  }
  __syncthreads();
  for (int i = tid; i < EM_W; i += 256) w_partials[(int64_t)blockIdx.x * EM_W + i] = red[i];
  if (tid < EM_C) b_partials[(int64_t)blockIdx.x * EM_C + tid] = redb[tid] + redb[EM_C + tid];
}

}  // namespace kvae
