// vae_conv_up.h — the decoder's two up-sampling blocks, Conv2d(32 -> 128, 3x3, pad 1) + PixelShuffle(2) + ReLU on 4x4
// and on 8x8 frames (reference kvae/vae/vae.py:92-101), as implicit GEMMs on the exact-f32 matrix cores with the
// WEIGHTS STATIONARY IN REGISTERS: the 128 x 288 weight matrix (147 KB) does not fit in LDS next to the frames, but a
// wave that owns 32 of the 128 output channels needs only its 32 x 288 slice = 144 registers per lane as the A operand
// of v_mfma_f32_32x32x2_f32.  One persistent workgroup (4 waves = 4 channel slices) per CU; an iteration is 128 input
// pixels (2 frames of 8x8 or 8 frames of 4x4: 16 KiB in, 64 KiB out), 576 MFMAs per wave; every B operand is one
// ds_read_b32 with a compile-time offset; bias, PixelShuffle and ReLU (forward) and the ReLU mask / un-shuffle
// (gradients) are fused, so the separate epilogue passes of these layers disappear.
// Library numbers at 12800 frames: Winograd forward 614 us / data gradient 547 us, implicit-GEMM weight gradient
// 842 us incl. layout transposes for the 8x8 layer (30 GMAC per pass: 385 us at the f32 matrix peak).
#pragma once
#include "vae_conv_mid.h"

namespace kvae {

constexpr int UP_CI = 32, UP_CO = 128, UP_W = UP_CO * UP_CI * 9;

template <int S>
struct UpDims {
  static constexpr int PF = S * S;                  // input pixels per frame (64 or 16)
  static constexpr int FPI = 128 / PF;              // frames per iteration (2 or 8)
  static constexpr int XFRAME = UP_CI * PF;         // input floats per frame
  static constexpr int IT_X = FPI * XFRAME;         // 4096 floats in per iteration
  static constexpr int YFRAME = UP_CO * PF;         // output floats per frame ([32, 2S, 2S])
  static constexpr int IT_Y = FPI * YFRAME;         // 16384 floats out per iteration
};

// ---------------------------------------------------------------------------------------------------------------
// forward: out[n, c, 2h+dy, 2w+dx] = relu(b[co] + sum_{ci,ky,kx} W[co,ci,ky,kx] x[n,ci,h+ky-1,w+kx-1]), co = 4c+2dy+dx.
// Wave wv owns co in [32 wv, 32 wv + 32); per pixel tile t (32 pixels): D[co][pixel] += A[co][k] B[k][pixel], k = (tap, ci).
// In the C/D register map the four registers r&3 of a lane are exactly the 2x2 shuffle block of its pixel.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_dec_up_fwd(const float *__restrict__ x, const float *__restrict__ W,
                                                    const float *__restrict__ bias, float *__restrict__ out, int64_t N) {
  using D = UpDims<S>;
  __shared__ float lds[D::IT_Y + D::IT_X];
  float *img = lds, *xin = lds + D::IT_Y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane & 31, half = lane >> 5;
  const int64_t iters = (N + D::FPI - 1) / D::FPI;
  const __amdgpu_buffer_rsrc_t rx = em_rsrc(x, N * D::XFRAME * 4), ry = em_rsrc(out, N * D::YFRAME * 4);
  const int rot = blockIdx.x & 15;

  float wreg[144];                                   // A[co = 32 wv + q][k = tap*32 + 2j + half]
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int j = 0; j < 16; ++j) wreg[tap * 16 + j] = W[((32 * wv + q) * UP_CI + 2 * j + half) * 9 + tap];
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = bias[32 * wv + KV_ACC_ROW(r, half)];

  float4 pre[4];
  int64_t it = blockIdx.x;
  auto fetch1 = [&](int64_t i, int j) { pre[j] = em_ld4(rx, (uint32_t)(i * D::IT_X + (tid + 256 * j) * 4) * 4u); };
  uint32_t done_base = 0x80000000u;                  // out of range: nothing to store yet
  auto flush1 = [&](int j) {
    const int c = (j + rot) & 15;
    em_st4(ry, done_base + (uint32_t)(tid + 256 * c) * 16u, reinterpret_cast<const float4 *>(img)[tid + 256 * c]);
  };
#pragma unroll
  for (int j = 0; j < 4; ++j) fetch1(it, j);
  for (int slot = 0; it < iters; it += gridDim.x, ++slot) {
    EM_STAMP(slot, 0);
    __syncthreads();                                 // previous iteration done with xin, its image complete
#pragma unroll
    for (int j = 0; j < 4; ++j) reinterpret_cast<float4 *>(xin)[tid + 256 * j] = pre[j];
    __syncthreads();
    EM_STAMP(slot, 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int gp = 32 * t + q, fl = gp / D::PF, pix = gp % D::PF, h = pix / S, w = pix % S;
      const int bbase = fl * D::XFRAME + half * D::PF + (h - 1) * S + (w - 1);
      int base[9];                                   // zero padding: border lanes read outside the LDS allocation (-> 0)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        const bool z = (ky == 0 && h == 0) || (ky == 2 && h == S - 1) || (kx == 0 && w == 0) || (kx == 2 && w == S - 1);
        base[tap] = (z ? EM_OOB : bbase) + ky * S + kx;
      }
      em_f16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float bw[36][4];
      auto rd = [&](int g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = g * 4 + u;
          bw[g][u] = xin[base[k / 16] + 2 * (k % 16) * D::PF];
        }
      };
      rd(0);
      rd(1);
#pragma unroll
      for (int g = 0; g < 36; ++g) {
        if (g + 2 < 36) rd(g + 2);
        // vector-memory instructions dealt one per MFMA group (see vae_conv_mid.h): the previous image out first
        if (t == 0 && g < 16) flush1(g);
        else if (t == 0 && g < 20) fetch1(it + gridDim.x, g - 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = KV_MFMA_F32(wreg[g * 4 + u], bw[g][u], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (t == 0) EM_STAMP(slot, 2);
      if (t == 3) EM_STAMP(slot, 4);
      if (t == 0) __syncthreads();                   // every wave has read the previous image out before it changes
      if (t == 0) EM_STAMP(slot, 3);
      float *o = img + fl * D::YFRAME + (2 * h) * (2 * S) + 2 * w;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 8 * wv + 2 * (r >> 2) + half, dy = (r & 3) >> 1, dx = r & 1;
        o[c * 4 * D::PF + dy * 2 * S + dx] = fmaxf(acc[r] + bv[r], 0.f);
      }
    }
    EM_STAMP(slot, 5);
    done_base = (uint32_t)(it * D::IT_Y) * 4u;       // frames >= N: dropped by the hardware
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 16; ++j) flush1(j);
}

// One float4 of the [32, 2S, 2S] gradient chunk -> four masked entries of the un-shuffled [128, S, S] LDS image.
template <int S, int CSTRIDE>
__device__ __forceinline__ void up_stage_gy(float *gyl, int e, const float4 g, const float4 o) {
  using D = UpDims<S>;
  const int f = e / D::YFRAME, c = (e % D::YFRAME) / (4 * D::PF), y = (e % (4 * D::PF)) / (2 * S), xx = e % (2 * S);
  float *d = gyl + (f * UP_CO + 4 * c + 2 * (y & 1)) * CSTRIDE + (y >> 1) * S + (xx >> 1);   // co = 4c + 2dy + dx
  d[0] = o.x > 0.f ? g.x : 0.f;             // dx = 0, w
  d[CSTRIDE] = o.y > 0.f ? g.y : 0.f;       // dx = 1, w
  d[1] = o.z > 0.f ? g.z : 0.f;             // dx = 0, w + 1
  d[CSTRIDE + 1] = o.w > 0.f ? g.w : 0.f;   // dx = 1, w + 1
}

// ---------------------------------------------------------------------------------------------------------------
// data gradient: g_x[n,ci,h,w] = sum_{co,ky,kx} W[co,ci,ky,kx] gy[n,co,h-ky+1,w-kx+1], gy = unshuffle(g_out * (out > 0)).
// The reduction over the 128 output channels is split over the four waves (32 each, their weight slice in registers);
// each wave produces D_w[ci][128 pixels] and the four partial results are folded through LDS.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_dec_up_bwd_data(const float *__restrict__ W, const float *__restrict__ out,
                                                         const float *__restrict__ g_out, float *__restrict__ g_x, int64_t N) {
  using D = UpDims<S>;
  __shared__ float lds[D::IT_Y + 4 * D::IT_X];
  float *gyl = lds, *part = lds + D::IT_Y;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane & 31, half = lane >> 5;
  const int64_t iters = (N + D::FPI - 1) / D::FPI;
  const __amdgpu_buffer_rsrc_t rg = em_rsrc(g_out, N * D::YFRAME * 4), ro = em_rsrc(out, N * D::YFRAME * 4),
                               rgx = em_rsrc(g_x, N * D::XFRAME * 4);
  float wreg[144];                                   // A[ci = q][k = tap*32 + col], co = 32 wv + col, col = 2j + half
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int j = 0; j < 16; ++j) wreg[tap * 16 + j] = W[((32 * wv + 2 * j + half) * UP_CI + q) * 9 + tap];

  float4 pg[16], po[16], res[4];
  int64_t it = blockIdx.x;
  auto fetch1 = [&](int64_t i, int j) {              // 32 pieces: 16 of g_out, then 16 of out
    const uint32_t e = (uint32_t)(i * D::IT_Y + (tid + 256 * (j & 15)) * 4) * 4u;
    if (j < 16) pg[j] = em_ld4(rg, e);
    else po[j - 16] = em_ld4(ro, e);
  };
  uint32_t done_base = 0x80000000u;
  auto flush1 = [&](int j) { em_st4(rgx, done_base + (uint32_t)(tid + 256 * j) * 16u, res[j]); };
#pragma unroll
  for (int j = 0; j < 32; ++j) fetch1(it, j);
  for (; it < iters; it += gridDim.x) {
    __syncthreads();                                 // previous iteration done with gyl and part
#pragma unroll
    for (int j = 0; j < 16; ++j) up_stage_gy<S, D::PF>(gyl, (tid + 256 * j) * 4, pg[j], po[j]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int gp = 32 * t + q, fl = gp / D::PF, pix = gp % D::PF, h = pix / S, w = pix % S;
      const int bbase = fl * D::YFRAME + (32 * wv + half) * D::PF + (h + 1) * S + (w + 1);
      int base[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        const bool z = (ky == 2 && h == 0) || (ky == 0 && h == S - 1) || (kx == 2 && w == 0) || (kx == 0 && w == S - 1);
        base[tap] = (z ? EM_OOB : bbase) - ky * S - kx;
      }
      em_f16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float bw[36][4];
      auto rd = [&](int g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = g * 4 + u;
          bw[g][u] = gyl[base[k / 16] + 2 * (k % 16) * D::PF];
        }
      };
      rd(0);
      rd(1);
#pragma unroll
      for (int g = 0; g < 36; ++g) {
        if (g + 2 < 36) rd(g + 2);
        const int G = t * 36 + g;                    // 4 result stores, then 32 prefetch loads, one every 4th group
        if (G % 4 == 0 && G / 4 < 4) flush1(G / 4);
        else if (G % 4 == 0 && G / 4 < 36) fetch1(it + gridDim.x, G / 4 - 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = KV_MFMA_F32(wreg[g * 4 + u], bw[g][u], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wv * D::IT_X + KV_ACC_ROW(r, half) * 128 + gp] = acc[r];
    }
    __syncthreads();
    // fold the four channel quarters; chunk order [frame][ci][pix] <- part[ci][frame*PF + pix]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = (tid + 256 * j) * 4, f = e / D::XFRAME, ci = (e % D::XFRAME) / D::PF, p = e % D::PF;
      const float *s0 = part + ci * 128 + f * D::PF + p;
      const float4 a = *reinterpret_cast<const float4 *>(s0), b = *reinterpret_cast<const float4 *>(s0 + D::IT_X),
                   c = *reinterpret_cast<const float4 *>(s0 + 2 * D::IT_X), d = *reinterpret_cast<const float4 *>(s0 + 3 * D::IT_X);
      res[j] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z),
                           (a.w + b.w) + (c.w + d.w));
    }
    done_base = (uint32_t)(it * D::IT_X) * 4u;       // frames >= N: dropped by the hardware
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) flush1(j);
}

// ---------------------------------------------------------------------------------------------------------------
// weight gradient: dW[co,ci,tap] = sum_{n,h,w} gy[n,co,h,w] x[n,ci,h+ky-1,w+kx-1]; db[co] = sum gy.
// Wave wv keeps the nine 32 x 32 tiles D_tap[co in its slice][ci] in 144 accumulator registers for the whole kernel;
// the 64 pixel pairs of an iteration are the K steps.  Channel planes are padded by one float in LDS (32 lanes on 32
// channels -> 32 banks).  Taps that fall off the top / bottom edge are skipped at compile time.
// ---------------------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void k_dec_up_wrw(const float *__restrict__ x, const float *__restrict__ out,
                                                    const float *__restrict__ g_out, float *__restrict__ w_partials,
                                                    float *__restrict__ b_partials, int64_t N) {
  using D = UpDims<S>;
  constexpr int CSP = D::PF + 1;
  __shared__ float lds[D::FPI * UP_CO * CSP + D::FPI * UP_CI * CSP];
  float *gyl = lds, *xin = lds + D::FPI * UP_CO * CSP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, q = lane & 31, half = lane >> 5;
  const int64_t iters = (N + D::FPI - 1) / D::FPI;
  const __amdgpu_buffer_rsrc_t rx = em_rsrc(x, N * D::XFRAME * 4), rg = em_rsrc(g_out, N * D::YFRAME * 4),
                               ro = em_rsrc(out, N * D::YFRAME * 4);
  const int abase = (32 * wv + q) * CSP + half;
  const int bbase = q * CSP + half - S - 1;
  const int x_l = half == 0 ? EM_OOB : bbase, x_r = half == 1 ? EM_OOB : bbase;   // column -1 / column S

  em_f16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  float4 px[4], pg[16], po[16];
  int64_t it = blockIdx.x;
  auto fetch1 = [&](int64_t i, int j) {              // 36 pieces: 4 of x, 16 of g_out, 16 of out
    if (j < 4) {
      px[j] = em_ld4(rx, (uint32_t)(i * D::IT_X + (tid + 256 * j) * 4) * 4u);
    } else {
      const uint32_t e = (uint32_t)(i * D::IT_Y + (tid + 256 * ((j - 4) & 15)) * 4) * 4u;
      if (j < 20) pg[j - 4] = em_ld4(rg, e);
      else po[j - 20] = em_ld4(ro, e);
    }
  };
#pragma unroll
  for (int j = 0; j < 36; ++j) fetch1(it, j);
  for (; it < iters; it += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = (tid + 256 * j) * 4, f = e / D::XFRAME, c = (e % D::XFRAME) / D::PF, p = e % D::PF;
      float *d = xin + (f * UP_CI + c) * CSP + p;
      d[0] = px[j].x; d[1] = px[j].y; d[2] = px[j].z; d[3] = px[j].w;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) up_stage_gy<S, CSP>(gyl, (tid + 256 * j) * 4, pg[j], po[j]);
    __syncthreads();
    // pixel pair i (pixels 2i, 2i+1 of the iteration's 128): its ten operands are read while pair i-1 is on the core
    float av[64], bw[64][9];
    auto rd = [&](int i) {
      const int fi = (2 * i) / D::PF, p0 = (2 * i) % D::PF, h = p0 / S, w0 = p0 % S;
      av[i] = gyl[abase + fi * UP_CO * CSP + p0];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        if ((ky == 0 && h == 0) || (ky == 2 && h == S - 1)) continue;          // off the top / bottom: no MFMA at all
        const bool zl = kx == 0 && w0 == 0, zr = kx == 2 && w0 == S - 2;
        bw[i][tap] = xin[(zl ? x_l : (zr ? x_r : bbase)) + fi * UP_CI * CSP + (h + ky) * S + w0 + kx];
      }
    };
    rd(0);
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int p0 = (2 * i) % D::PF, h = p0 / S;
      if (i + 1 < 64) rd(i + 1);
      if (i < 36) fetch1(it + gridDim.x, i);          // next iteration's loads, one per group (see vae_conv_mid.h)
      __builtin_amdgcn_sched_barrier(0);
      bsum += av[i];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3;
        if ((ky == 0 && h == 0) || (ky == 2 && h == S - 1)) continue;
        acc[tap] = KV_MFMA_F32(av[i], bw[i][tap], acc[tap]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float *wp = w_partials + (int64_t)blockIdx.x * UP_W;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) wp[((32 * wv + KV_ACC_ROW(r, half)) * UP_CI + q) * 9 + tap] = acc[tap][r];
  bsum += __shfl_xor(bsum, 32);
  if (half == 0) b_partials[(int64_t)blockIdx.x * UP_CO + 32 * wv + q] = bsum;
}

}  // namespace kvae
