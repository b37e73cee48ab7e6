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
  __shared__ float lds[EM_W + D::IT_IN];
  float *Wl = lds, *fr = lds + EM_W;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane & 31, half = lane >> 5;
  em_load_weights(Wl, W, true);
  const int64_t iters = (N + D::FPI - 1) / D::FPI, total = N * D::FRAME;
  const int gp = 32 * wv + q, fl = gp / D::PF, pix = gp % D::PF, oh = pix / D::OS, ow = pix % D::OS;
  const bool top = oh == 0, left = ow == 0;
  const int bbase = fl * D::FRAME + half * D::PLANE + (2 * oh - 1) * S + (2 * ow - 1);
  const int abase = half * 32 + q;
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias[KV_ACC_ROW(r, half)];

  float4 pre[16];
  int64_t it = blockIdx.x;
  auto fetch = [&](int64_t i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t e = i * D::IT_IN + (int64_t)(tid + 256 * j) * 4;
      pre[j] = e < total ? *reinterpret_cast<const float4 *>(in + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (it < iters) fetch(it);
  for (; it < iters; it += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) reinterpret_cast<float4 *>(fr)[tid + 256 * j] = pre[j];
    __syncthreads();
    if (it + gridDim.x < iters) fetch(it + gridDim.x);
    em_f16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      const bool zero = (ky == 0 && top) || (kx == 0 && left);
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) {
        const float a = Wl[abase + (tap * 32 + 2 * jj) * 32];
        float b = fr[bbase + ky * S + kx + 2 * jj * D::PLANE];
        b = zero ? 0.f : b;
        acc = KV_MFMA_F32(a, b, acc);
      }
    }
    const int64_t frame = it * D::FPI + fl;
    if (frame < N) {
      float *o = out + frame * EM_C * D::PF + pix;
#pragma unroll
      for (int r = 0; r < 16; ++r) o[KV_ACC_ROW(r, half) * D::PF] = fmaxf(acc[r] + bv[r], 0.f);
    }
  }
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
  const int abase = half * 32 + q;
  const int obase = fl * D::FRAME + (2 * a_) * S + 2 * b_;

  float4 pg[4], po[4];
  int64_t it = blockIdx.x;
  auto fetch = [&](int64_t i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = i * D::IT_OUT + (int64_t)(tid + 256 * j) * 4;
      const bool ok = e < total_out;
      pg[j] = ok ? *reinterpret_cast<const float4 *>(g_out + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      po[j] = ok ? *reinterpret_cast<const float4 *>(out + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (it < iters) fetch(it);
  for (; it < iters; it += gridDim.x) {
    __syncthreads();                                   // previous iteration's LDS images fully consumed
#pragma unroll
    for (int j = 0; j < 4; ++j)
      reinterpret_cast<float4 *>(gm)[tid + 256 * j] = make_float4(po[j].x > 0.f ? pg[j].x : 0.f, po[j].y > 0.f ? pg[j].y : 0.f,
                                                                   po[j].z > 0.f ? pg[j].z : 0.f, po[j].w > 0.f ? pg[j].w : 0.f);
    __syncthreads();
    if (it + gridDim.x < iters) fetch(it + gridDim.x);
#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
      const int ph = cls >> 1, pw = cls & 1;
      em_f16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        if ((ky & 1) == ph) continue;                  // row parity: even rows take ky = 1, odd rows ky = 0 and 2
        const int dy = (ph == 1 && ky == 0) ? 1 : 0;   // oh = a + dy
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if ((kx & 1) == pw) continue;
          const int dx = (pw == 1 && kx == 0) ? 1 : 0;
          const int tap = ky * 3 + kx;
          const bool zero = (dy == 1 && bottom) || (dx == 1 && right);
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) {
            const float a = Wl[abase + (tap * 32 + 2 * jj) * 32];
            float b = gm[gbase + 2 * jj * D::PF + dy * D::OS + dx];
            b = zero ? 0.f : b;
            acc = KV_MFMA_F32(a, b, acc);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[obase + KV_ACC_ROW(r, half) * D::PLANE + ph * S + pw] = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t e = it * D::IT_IN + (int64_t)(tid + 256 * j) * 4;
      if (e < total_in) *reinterpret_cast<float4 *>(g_in + e) = reinterpret_cast<const float4 *>(ot)[tid + 256 * j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co,ci,tap] = sum_{n,oh,ow} gm[n,co,oh,ow] in[n,ci,2oh+ky-1,2ow+kx-1]; db[co] = sum gm.
// GEMM view: nine 32 x 32 tiles D_tap[co][ci] += A[co][pixel] B_tap[pixel][ci], 819k pixels deep.  The 64 pixel pairs
// of an iteration are split over the four waves (16 each, x 9 taps = 144 MFMAs); every wave keeps its own nine tiles
// in 144 accumulator registers for the whole kernel and writes them out once as a partial row.
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

  em_f16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  float4 px[16], pg[4], po[4];
  int64_t it = blockIdx.x;
  auto fetch = [&](int64_t i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int64_t e = i * D::IT_IN + (int64_t)(tid + 256 * j) * 4;
      px[j] = e < total_in ? *reinterpret_cast<const float4 *>(in + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t e = i * D::IT_OUT + (int64_t)(tid + 256 * j) * 4;
      const bool ok = e < total_out;
      pg[j] = ok ? *reinterpret_cast<const float4 *>(g_out + e) : make_float4(0.f, 0.f, 0.f, 0.f);
      po[j] = ok ? *reinterpret_cast<const float4 *>(out + e) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (it < iters) fetch(it);
  for (; it < iters; it += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int e = (tid + 256 * j) * 4, f = e / D::FRAME, c = (e % D::FRAME) / D::PLANE, p = e % D::PLANE;
      float *d = xin + (f * EM_C + c) * CSP + p;
      d[0] = px[j].x; d[1] = px[j].y; d[2] = px[j].z; d[3] = px[j].w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = (tid + 256 * j) * 4, f = e / (EM_C * D::PF), c = (e / D::PF) % EM_C, p = e % D::PF;
      float *d = gm + (f * EM_C + c) * GSP + p;
      d[0] = po[j].x > 0.f ? pg[j].x : 0.f; d[1] = po[j].y > 0.f ? pg[j].y : 0.f;
      d[2] = po[j].z > 0.f ? pg[j].z : 0.f; d[3] = po[j].w > 0.f ? pg[j].w : 0.f;
    }
    __syncthreads();
    if (it + gridDim.x < iters) fetch(it + gridDim.x);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // pixel pair i of this wave: frame offset fi, first pixel p (even) relative to wpix0
      const int fi = S == 16 ? 0 : i / 8, p = S == 16 ? 2 * i : 2 * (i % 8);
      const int ohr = p / D::OS, owr = p % D::OS;          // relative row, column of the pair's first pixel
      const float a = gm[abase + fi * EM_C * GSP + p];
      bsum += a;
      const bool ztop = (oh0 + ohr) == 0;                   // wave-uniform
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        float b = xin[bbase + fi * EM_C * CSP + (2 * ohr + ky) * S + 2 * owr + kx];
        if (ky == 0) b = ztop ? 0.f : b;
        if (kx == 0 && owr == 0) b = half == 0 ? 0.f : b;   // column -1 belongs to the pair's first pixel only
        acc[tap] = KV_MFMA_F32(a, b, acc[tap]);
      }
    }
  }
  // D_tap[co][ci]: column ci = q on the lane, rows co in the registers
  float *wp = w_partials + (int64_t)(blockIdx.x * 4 + wv) * EM_W;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) wp[(KV_ACC_ROW(r, half) * EM_C + q) * 9 + tap] = acc[tap][r];
  b_partials[(int64_t)((blockIdx.x * 4 + wv) * 2 + half) * EM_C + q] = bsum;
}

}  // namespace kvae
