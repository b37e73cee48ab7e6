// vae_conv_edge.h — direct convolutions for the two DEGENERATE layers of the frame VAE, where an implicit-GEMM library
// kernel has almost no channel dimension to tile over (measured with MIOpen on MI355X at 12800 frames, fp32):
//   * decoder head  Conv2d(32 -> 4, 3x3, pad 1) + PixelShuffle(2) on 16x16   (reference vae.py:103-104):
//       forward 624 us, data-gradient 255 us, weight-gradient 1000 us for 3.8 GMAC each;
//   * encoder stem  Conv2d(1 -> 32, 3x3, stride 2, pad 1) + ReLU on 32x32      (reference vae.py:20-31):
//       forward 244 us, weight-gradient 386 us for 0.9 GMAC each.
// Both are HBM-bound by their big side (a 419 MB activation), so: one workgroup per frame, the frame tile staged in LDS
// with a zero halo, every thread owns one output pixel (or one input channel for the weight gradients) and keeps all
// its accumulators in registers; weights are wave-uniform and travel through scalar loads.  Bias, PixelShuffle, ReLU
// and the ReLU mask of the backward are fused into the same passes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {

// ------------------------------------------------------------------------------------------------------------------
// decoder head: in [N,32,16,16] -> logits [N,1,32,32] = pixel_shuffle_2(conv3x3(in, W[4,32,3,3]) + b[4])
// ------------------------------------------------------------------------------------------------------------------
constexpr int DH_CI = 32, DH_CO = 4, DH_S = 16, DH_TS = 18, DH_CS = 325;   // tile side with halo, padded channel stride

__device__ __forceinline__ void dh_load_tile(float *tile, const float *__restrict__ src) {
  for (int i = threadIdx.x; i < DH_CI * DH_CS; i += 256) tile[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < DH_CI * DH_S * DH_S / 4; i += 256) {
    const float4 v = reinterpret_cast<const float4 *>(src)[i];
    const int e = i * 4, ci = e >> 8, hw = e & 255, h = hw >> 4, w = hw & 15;
    float *d = tile + ci * DH_CS + (h + 1) * DH_TS + (w + 1);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

__global__ __launch_bounds__(256) void k_dec_head_fwd(const float *__restrict__ in, const float *__restrict__ W,
                                                      const float *__restrict__ bias, float *__restrict__ logits) {
  __shared__ float tile[DH_CI * DH_CS];
  const int64_t n = blockIdx.x;
  dh_load_tile(tile, in + n * DH_CI * 256);
  __syncthreads();
  const int h = threadIdx.x >> 4, w = threadIdx.x & 15;
  float acc[DH_CO];
#pragma unroll
  for (int co = 0; co < DH_CO; ++co) acc[co] = bias[co];
#pragma unroll 4
  for (int ci = 0; ci < DH_CI; ++ci) {
    const float *t = tile + ci * DH_CS + h * DH_TS + w;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float v = t[ky * DH_TS + kx];
#pragma unroll
        for (int co = 0; co < DH_CO; ++co) acc[co] = fmaf(W[((co * DH_CI + ci) * 3 + ky) * 3 + kx], v, acc[co]);
      }
  }
  float *o = logits + n * 1024 + (2 * h) * 32 + 2 * w;          // co = 2*dy + dx -> pixel (2h+dy, 2w+dx)
  *reinterpret_cast<float2 *>(o) = make_float2(acc[0], acc[1]);
  *reinterpret_cast<float2 *>(o + 32) = make_float2(acc[2], acc[3]);
}

// g_in[n,ci,h,w] = sum_{co,ky,kx} W[co,ci,ky,kx] g_conv[n,co,h-ky+1,w-kx+1],  g_conv = pixel_unshuffle(g_logits)
__global__ __launch_bounds__(256) void k_dec_head_bwd_data(const float *__restrict__ g_logits, const float *__restrict__ W,
                                                           float *__restrict__ g_in) {
  __shared__ float gt[DH_CO * DH_CS];
  const int64_t n = blockIdx.x;
  for (int i = threadIdx.x; i < DH_CO * DH_CS; i += 256) gt[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int oh = i >> 5, ow = i & 31, co = (oh & 1) * 2 + (ow & 1);
    gt[co * DH_CS + ((oh >> 1) + 1) * DH_TS + (ow >> 1) + 1] = g_logits[n * 1024 + i];
  }
  __syncthreads();
  const int h = threadIdx.x >> 4, w = threadIdx.x & 15;
  float g[DH_CO][9];
#pragma unroll
  for (int co = 0; co < DH_CO; ++co)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) g[co][ky * 3 + kx] = gt[co * DH_CS + (h + 2 - ky) * DH_TS + (w + 2 - kx)];
  float *o = g_in + n * DH_CI * 256 + threadIdx.x;
#pragma unroll 4
  for (int ci = 0; ci < DH_CI; ++ci) {
    float acc = 0.f;
#pragma unroll
    for (int co = 0; co < DH_CO; ++co)
#pragma unroll
      for (int k = 0; k < 9; ++k) acc = fmaf(W[(co * DH_CI + ci) * 9 + k], g[co][k], acc);
    o[ci * 256] = acc;
  }
}

// partial[blk, co, ci, k] = sum over this block's frames of sum_{h,w} g_conv[n,co,h,w] in[n,ci,h+ky-1,w+kx-1];
// partial_b[blk, co] = sum g_conv.  Thread = (ci = tid & 31, pixel group = tid >> 5: 32 pixels each).
__global__ __launch_bounds__(256) void k_dec_head_wrw(const float *__restrict__ in, const float *__restrict__ g_logits,
                                                      float *__restrict__ partial, float *__restrict__ partial_b, int64_t N) {
  __shared__ float tile[DH_CI * DH_CS];
  __shared__ float gt[DH_CO * 256];
  const int ci = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float acc[DH_CO][9], accb[DH_CO];
#pragma unroll
  for (int co = 0; co < DH_CO; ++co) {
    accb[co] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[co][k] = 0.f;
  }
  for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
    __syncthreads();                                  // previous frame's tile fully consumed
    dh_load_tile(tile, in + n * DH_CI * 256);
    for (int i = threadIdx.x; i < 1024; i += 256) {
      const int oh = i >> 5, ow = i & 31, co = (oh & 1) * 2 + (ow & 1);
      gt[co * 256 + (oh >> 1) * 16 + (ow >> 1)] = g_logits[n * 1024 + i];
    }
    __syncthreads();
    for (int i = 0; i < 32; ++i) {
      const int hw = grp * 32 + i, h = hw >> 4, w = hw & 15;
      float gv[DH_CO];
#pragma unroll
      for (int co = 0; co < DH_CO; ++co) { gv[co] = gt[co * 256 + hw]; accb[co] += gv[co]; }
      const float *t = tile + ci * DH_CS + h * DH_TS + w;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float v = t[ky * DH_TS + kx];
#pragma unroll
          for (int co = 0; co < DH_CO; ++co) acc[co][ky * 3 + kx] = fmaf(gv[co], v, acc[co][ky * 3 + kx]);
        }
    }
  }
  __syncthreads();
  // reduce the 8 pixel groups through LDS (reuse tile: 8 * 32 * 36 floats = 9216 <= 10400)
  float *red = tile;
#pragma unroll
  for (int co = 0; co < DH_CO; ++co)
#pragma unroll
    for (int k = 0; k < 9; ++k) red[(grp * 32 + ci) * 36 + co * 9 + k] = acc[co][k];
  if (ci == 0) {
#pragma unroll
    for (int co = 0; co < DH_CO; ++co) gt[grp * 4 + co] = accb[co];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < DH_CO * DH_CI * 9; o += 256) {
    const int co = o / (DH_CI * 9), r = o - co * DH_CI * 9, c2 = r / 9, k = r - c2 * 9;
    float s = 0.f;
    for (int gq = 0; gq < 8; ++gq) s += red[(gq * 32 + c2) * 36 + co * 9 + k];
    partial[(int64_t)blockIdx.x * (DH_CO * DH_CI * 9) + o] = s;
  }
  if (threadIdx.x < DH_CO) {
    float s = 0.f;
    for (int gq = 0; gq < 8; ++gq) s += gt[gq * 4 + threadIdx.x];
    partial_b[(int64_t)blockIdx.x * DH_CO + threadIdx.x] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// encoder stem: x [N,1,32,32] -> out [N,32,16,16] = relu(conv3x3_stride2_pad1(x, W[32,1,3,3]) + b[32])
// ------------------------------------------------------------------------------------------------------------------
constexpr int ES_CO = 32, ES_IN = 32, ES_OUT = 16, ES_TS = 35;   // input tile 34 x 34 (halo), row stride 35

__device__ __forceinline__ void es_load_tile(float *tile, const float *__restrict__ x) {
  for (int i = threadIdx.x; i < 34 * ES_TS; i += 256) tile[i] = 0.f;
  __syncthreads();
  const float4 v = reinterpret_cast<const float4 *>(x)[threadIdx.x];
  const int e = threadIdx.x * 4, h = e >> 5, w = e & 31;
  float *d = tile + (h + 1) * ES_TS + (w + 1);
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

__global__ __launch_bounds__(256) void k_enc_stem_fwd(const float *__restrict__ x, const float *__restrict__ W,
                                                      const float *__restrict__ bias, float *__restrict__ out) {
  __shared__ float tile[34 * ES_TS];
  const int64_t n = blockIdx.x;
  es_load_tile(tile, x + n * 1024);
  __syncthreads();
  const int oh = threadIdx.x >> 4, ow = threadIdx.x & 15;
  float v[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) v[ky * 3 + kx] = tile[(2 * oh + ky) * ES_TS + 2 * ow + kx];
  float *o = out + n * ES_CO * 256 + threadIdx.x;
#pragma unroll 8
  for (int co = 0; co < ES_CO; ++co) {
    float acc = bias[co];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc = fmaf(W[co * 9 + k], v[k], acc);
    o[co * 256] = fmaxf(acc, 0.f);
  }
}

// partial[blk, co, k] = sum_frames sum_{oh,ow} gm[n,co,oh,ow] x[n, 2oh+ky-1, 2ow+kx-1], gm = g_out * (out > 0);
// partial_b[blk, co] = sum gm.   Thread = (co = tid & 31, pixel group = tid >> 5).
__global__ __launch_bounds__(256) void k_enc_stem_wrw(const float *__restrict__ x, const float *__restrict__ out,
                                                      const float *__restrict__ g_out, float *__restrict__ partial,
                                                      float *__restrict__ partial_b, int64_t N) {
  __shared__ float tile[34 * ES_TS];
  __shared__ float gt[ES_CO * 257];
  const int co = threadIdx.x & 31, grp = threadIdx.x >> 5;
  float acc[9], accb = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0.f;
  for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
    __syncthreads();
    es_load_tile(tile, x + n * 1024);
    const float4 *g4 = reinterpret_cast<const float4 *>(g_out + n * ES_CO * 256);
    const float4 *o4 = reinterpret_cast<const float4 *>(out + n * ES_CO * 256);
    for (int i = threadIdx.x; i < ES_CO * 64; i += 256) {
      const float4 gv = g4[i], ov = o4[i];
      const int e = i * 4, c = e >> 8, hw = e & 255;
      float *d = gt + c * 257 + hw;
      d[0] = ov.x > 0.f ? gv.x : 0.f; d[1] = ov.y > 0.f ? gv.y : 0.f;
      d[2] = ov.z > 0.f ? gv.z : 0.f; d[3] = ov.w > 0.f ? gv.w : 0.f;
    }
    __syncthreads();
    for (int i = 0; i < 32; ++i) {
      const int hw = grp * 32 + i, oh = hw >> 4, ow = hw & 15;
      const float gv = gt[co * 257 + hw];
      accb += gv;
      const float *t = tile + (2 * oh) * ES_TS + 2 * ow;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = fmaf(gv, t[ky * ES_TS + kx], acc[ky * 3 + kx]);
    }
  }
  __syncthreads();
  float *red = gt;   // 8 groups x 32 channels x 10
#pragma unroll
  for (int k = 0; k < 9; ++k) red[(grp * 32 + co) * 10 + k] = acc[k];
  red[(grp * 32 + co) * 10 + 9] = accb;
  __syncthreads();
  for (int o = threadIdx.x; o < ES_CO * 10; o += 256) {
    const int c = o / 10, k = o - c * 10;
    float s = 0.f;
    for (int gq = 0; gq < 8; ++gq) s += red[(gq * 32 + c) * 10 + k];
    if (k < 9) partial[(int64_t)blockIdx.x * (ES_CO * 9) + c * 9 + k] = s;
    else partial_b[(int64_t)blockIdx.x * ES_CO + c] = s;
  }
}

}  // namespace kvae
