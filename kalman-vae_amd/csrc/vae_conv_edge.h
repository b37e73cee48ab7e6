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
constexpr int DH_W = DH_CO * DH_CI * 9;                                    // 1152 weights
typedef float kv_f2 __attribute__((ext_vector_type(2)));
typedef float kv_f4 __attribute__((ext_vector_type(4)));

// Weights re-laid for wide scalar loads and packed FMAs: scratch[0:1152] = Wf[ci][k][co] (forward: the 4 output
// channels of one tap are one s_load_dwordx4), scratch[1152:2304] = Wb[co][k][ci] (data gradient: the 32 input
// channels of one tap are two s_load_dwordx16).
__global__ __launch_bounds__(256) void k_dec_head_prep(const float *__restrict__ W, float *__restrict__ scratch) {
  for (int i = threadIdx.x; i < DH_W; i += 256) {
    const int co = i / (DH_CI * 9), r = i - co * DH_CI * 9, ci = r / 9, k = r - ci * 9;
    const float w = W[i];
    scratch[(ci * 9 + k) * DH_CO + co] = w;
    scratch[DH_W + (co * 9 + k) * DH_CI + ci] = w;
  }
}

__device__ __forceinline__ void dh_zero_halo(float *tile, int channels) {
  for (int i = threadIdx.x; i < channels * 68; i += 256) {
    const int c = i / 68, j = i - c * 68;
    int r, q;
    if (j < 18) { r = 0; q = j; }
    else if (j < 36) { r = 17; q = j - 18; }
    else if (j < 52) { r = j - 35; q = 0; }
    else { r = j - 51; q = 17; }
    tile[c * DH_CS + r * DH_TS + q] = 0.f;
  }
}
__device__ __forceinline__ void dh_store_interior(float *tile, int i, const float4 v) {   // i = float4 index in [32,16,16]
  const int e = i * 4, ci = e >> 8, hw = e & 255, h = hw >> 4, w = hw & 15;
  float *d = tile + ci * DH_CS + (h + 1) * DH_TS + (w + 1);
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}

// The frame sits in LDS as a plain 32 x 16 x 16 copy (32 KiB: five workgroups per CU instead of three with a halo, and
// the staging is eight ds_write_b128 per thread).  The zero padding comes from the out-of-range trick of vae_conv_mid.h:
// border lanes take a base far outside the LDS allocation for the taps that fall off the frame, where DS reads return 0.
__global__ __launch_bounds__(256) void k_dec_head_fwd(const float *__restrict__ in, const float *__restrict__ scratch,
                                                      const float *__restrict__ bias, float *__restrict__ logits) {
  __shared__ float tile[DH_CI * 256];
  const int64_t n = blockIdx.x;
  const float4 *src = reinterpret_cast<const float4 *>(in + n * DH_CI * 256);
  float4 pre[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) pre[j] = src[threadIdx.x + 256 * j];
#pragma unroll
  for (int j = 0; j < 8; ++j) reinterpret_cast<float4 *>(tile)[threadIdx.x + 256 * j] = pre[j];
  __syncthreads();
  const int h = threadIdx.x >> 4, w = threadIdx.x & 15;
  constexpr int OOB = 1 << 24;                       // float index 64 MiB past the start of LDS
  const int tb = (h - 1) * 16 + (w - 1);
  int base[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int ky = tap / 3, kx = tap % 3;
    const bool z = (ky == 0 && h == 0) || (ky == 2 && h == 15) || (kx == 0 && w == 0) || (kx == 2 && w == 15);
    base[tap] = (z ? OOB : tb) + ky * 16 + kx;
  }
  kv_f2 a01 = {bias[0], bias[1]}, a23 = {bias[2], bias[3]};
  const kv_f4 *Wf = reinterpret_cast<const kv_f4 *>(scratch);
#pragma unroll 4
  for (int ci = 0; ci < DH_CI; ++ci) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float v = tile[base[tap] + ci * 256];
      const kv_f4 wv = Wf[ci * 9 + tap];
      a01 += wv.xy * v;
      a23 += wv.zw * v;
    }
  }
  float *o = logits + n * 1024 + (2 * h) * 32 + 2 * w;          // co = 2*dy + dx -> pixel (2h+dy, 2w+dx)
  *reinterpret_cast<float2 *>(o) = make_float2(a01.x, a01.y);
  *reinterpret_cast<float2 *>(o + 32) = make_float2(a23.x, a23.y);
}

// g_in[n,ci,h,w] = sum_{co,ky,kx} W[co,ci,ky,kx] g_conv[n,co,h-ky+1,w-kx+1],  g_conv = pixel_unshuffle(g_logits)
__global__ __launch_bounds__(256) void k_dec_head_bwd_data(const float *__restrict__ g_logits,
                                                           const float *__restrict__ scratch, float *__restrict__ g_in) {
  __shared__ float gt[DH_CO * DH_CS];
  const int64_t n = blockIdx.x;
  float gp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) gp[j] = g_logits[n * 1024 + threadIdx.x + 256 * j];
  dh_zero_halo(gt, DH_CO);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = threadIdx.x + 256 * j, oh = i >> 5, ow = i & 31, co = (oh & 1) * 2 + (ow & 1);
    gt[co * DH_CS + ((oh >> 1) + 1) * DH_TS + (ow >> 1) + 1] = gp[j];
  }
  __syncthreads();
  const int h = threadIdx.x >> 4, w = threadIdx.x & 15;
  kv_f2 acc[DH_CI / 2];
#pragma unroll
  for (int j = 0; j < DH_CI / 2; ++j) acc[j] = kv_f2{0.f, 0.f};
  const kv_f2 *Wb = reinterpret_cast<const kv_f2 *>(scratch + DH_W);
#pragma unroll
  for (int co = 0; co < DH_CO; ++co)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float v = gt[co * DH_CS + (h + 2 - ky) * DH_TS + (w + 2 - kx)];
        const kv_f2 *wr = Wb + (co * 9 + ky * 3 + kx) * (DH_CI / 2);
#pragma unroll
        for (int j = 0; j < DH_CI / 2; ++j) acc[j] += wr[j] * v;
      }
  float *o = g_in + n * DH_CI * 256 + threadIdx.x;
#pragma unroll
  for (int j = 0; j < DH_CI / 2; ++j) {
    o[(2 * j) * 256] = acc[j].x;
    o[(2 * j + 1) * 256] = acc[j].y;
  }
}

// partial[blk, co, ci, k] = sum over this block's frames of sum_{h,w} g_conv[n,co,h,w] in[n,ci,h+ky-1,w+kx-1];
// partial_b[blk, co] = sum g_conv.  Thread = (ci = tid & 31, pixel group = tid >> 5: 32 pixels each).  The next
// frame is fetched into registers while the current one is being reduced out of LDS.
__global__ __launch_bounds__(256) void k_dec_head_wrw(const float *__restrict__ in, const float *__restrict__ g_logits,
                                                      float *__restrict__ partial, float *__restrict__ partial_b, int64_t N) {
  // frame copy with the channel planes padded by one float (32 lanes on 32 channels -> 32 banks), no halo: the rows
  // above / below the frame are read from outside the LDS allocation (-> 0), the columns left / right of it are
  // compile-time zeros of the sliding window.  9248 floats = 37 KiB: four workgroups per CU.
  constexpr int CS = 257, OOB = 1 << 24;
  __shared__ float lds[DH_CI * CS + DH_CO * 256];
  __shared__ float redb[8 * DH_CO];
  float *tile = lds, *gt = lds + DH_CI * CS;
  const int ci = threadIdx.x & 31, grp = threadIdx.x >> 5;
  // which 16 bytes of the frame a thread stages: sixteen lanes per channel plane, four planes per wave - consecutive lanes
  // inside one 256-float plane make the four scalar writes per 16 bytes (planes are padded by one float) 8-way bank conflicts
  const int tin = (((threadIdx.x & 63) >> 4) << 6) | ((threadIdx.x >> 6) << 4) | (threadIdx.x & 15);
  kv_f2 acc[DH_CO / 2][9];
  float accb[DH_CO];
#pragma unroll
  for (int c = 0; c < DH_CO / 2; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[c][k] = kv_f2{0.f, 0.f};
#pragma unroll
  for (int co = 0; co < DH_CO; ++co) accb[co] = 0.f;
  float4 pre[8];
  float gp[4];
  int64_t n = blockIdx.x;
  if (n < N) {
    const float4 *src = reinterpret_cast<const float4 *>(in + n * DH_CI * 256);
#pragma unroll
    for (int j = 0; j < 8; ++j) pre[j] = src[tin + 256 * j];
#pragma unroll
    for (int j = 0; j < 4; ++j) gp[j] = g_logits[n * 1024 + threadIdx.x + 256 * j];
  }
  for (; n < N; n += gridDim.x) {
    __syncthreads();                                  // previous frame's tile fully consumed
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = (tin + 256 * j) * 4;
      float *d = tile + (e >> 8) * CS + (e & 255);
      d[0] = pre[j].x; d[1] = pre[j].y; d[2] = pre[j].z; d[3] = pre[j].w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = threadIdx.x + 256 * j, oh = i >> 5, ow = i & 31, co = (oh & 1) * 2 + (ow & 1);
      gt[co * 256 + (oh >> 1) * 16 + (ow >> 1)] = gp[j];
    }
    __syncthreads();
    const int64_t nn = n + gridDim.x;
    if (nn < N) {
      const float4 *src = reinterpret_cast<const float4 *>(in + nn * DH_CI * 256);
#pragma unroll
      for (int j = 0; j < 8; ++j) pre[j] = src[tin + 256 * j];
#pragma unroll
      for (int j = 0; j < 4; ++j) gp[j] = g_logits[nn * 1024 + threadIdx.x + 256 * j];
    }
    // this thread's 32 pixels are rows 2*grp and 2*grp+1; a 3x3 window slides along each row, so a pixel costs three
    // new input reads (its right-hand column) instead of nine
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {                  // fully unrolled on purpose: 234 VGPRs / 2 workgroups per CU beat
      const int h = 2 * grp + rr;                     // the 128-VGPR / 4-per-CU build (148 vs 175 us)
      int rb[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const bool off = (ky == 0 && rr == 0 && grp == 0) || (ky == 2 && rr == 1 && grp == 7);
        rb[ky] = (off ? OOB : ci * CS) + (h + ky - 1) * 16;
      }
      float c0[3] = {0.f, 0.f, 0.f}, c1[3], c2[3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) c1[ky] = tile[rb[ky]];
#pragma unroll
      for (int w = 0; w < 16; ++w) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) c2[ky] = w + 1 < 16 ? tile[rb[ky] + w + 1] : 0.f;
        const int hw = h * 16 + w;
        const kv_f2 g01 = {gt[hw], gt[256 + hw]}, g23 = {gt[512 + hw], gt[768 + hw]};
        accb[0] += g01.x; accb[1] += g01.y; accb[2] += g23.x; accb[3] += g23.y;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          acc[0][ky * 3 + 0] += g01 * c0[ky]; acc[1][ky * 3 + 0] += g23 * c0[ky];
          acc[0][ky * 3 + 1] += g01 * c1[ky]; acc[1][ky * 3 + 1] += g23 * c1[ky];
          acc[0][ky * 3 + 2] += g01 * c2[ky]; acc[1][ky * 3 + 2] += g23 * c2[ky];
          c0[ky] = c1[ky];
          c1[ky] = c2[ky];
        }
      }
    }
  }
  __syncthreads();
  // reduce the 8 pixel groups through LDS (reuse the frame buffers: 8 * 32 * 36 floats = 9216 <= 9248)
  float *red = lds;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float *r = red + (grp * 32 + ci) * 36 + k;
    r[0] = acc[0][k].x; r[9] = acc[0][k].y; r[18] = acc[1][k].x; r[27] = acc[1][k].y;
  }
  if (ci == 0) {
#pragma unroll
    for (int co = 0; co < DH_CO; ++co) redb[grp * 4 + co] = accb[co];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < DH_W; o += 256) {
    const int co = o / (DH_CI * 9), r = o - co * DH_CI * 9, c2 = r / 9, k = r - c2 * 9;
    float s = 0.f;
    for (int gq = 0; gq < 8; ++gq) s += red[(gq * 32 + c2) * 36 + co * 9 + k];
    partial[(int64_t)blockIdx.x * DH_W + o] = s;
  }
  if (threadIdx.x < DH_CO) {
    float s = 0.f;
    for (int gq = 0; gq < 8; ++gq) s += redb[gq * 4 + threadIdx.x];
    partial_b[(int64_t)blockIdx.x * DH_CO + threadIdx.x] = s;
  }
}

// (An f32-MFMA version of this weight gradient - D[ci 32][(co, tap) 36 of 48] += A[ci][4 pixels] B[4 pixels][(co, tap)], the
//  gradient read at the tap-shifted pixel from zero-bordered planes, 24 accumulator registers, four workgroups per CU - was
//  written and measured in round 2: 143 us against 153 alone, no difference on the training step; not kept.)

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
                                                      const float *__restrict__ bias, float *__restrict__ out,
                                                      uint32_t *__restrict__ relu_bits) {
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
  uint32_t bits = 0;                                  // the thread walks all 32 channels of its pixel: their ReLU mask is one word
#pragma unroll 8
  for (int co = 0; co < ES_CO; ++co) {
    float acc = bias[co];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc = fmaf(W[co * 9 + k], v[k], acc);
    o[co * 256] = fmaxf(acc, 0.f);
    bits |= (acc > 0.f ? 1u : 0u) << co;
  }
  if (relu_bits) relu_bits[n * 256 + threadIdx.x] = bits;
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

// The same weight gradient on the f32 matrix cores: per frame dW[32 co][9 taps] = Gm[32 x 256 pixels] Xp[256 pixels x 9 taps] is
// 128 v_mfma_f32_16x16x4_f32 (two 16-channel halves x 64 k-steps of four pixels; the 16 tap columns carry the 9 taps, a column
// of ones - which makes the bias gradient fall out of the same product - and six ignored ones).  The VALU version above spends
// ten LDS reads and nine FMAs per (channel, pixel) and runs at 2.9 TB/s; here every MFMA costs one LDS read per operand and the
// kernel is left with its 0.47 GB of g_out / out / x traffic.  Wave wv takes pixels 64 wv .. 64 wv + 63 of every frame.
typedef float es_f4 __attribute__((ext_vector_type(4)));
// BITS: the ReLU mask comes as one 32-bit word per pixel (written by the forward: 1 KB per frame instead of the 32 KB of `out`,
// 0.42 of this kernel's 0.89 GB).  What the parts cost at 12800 frames (tools/stem_wrw_diag.hip, compiled out one at a time):
// streaming g_out alone 70 us (6.7 TB/s), + LDS staging 9, + frame tile 29 (loaded synchronously: now prefetched), + MFMAs 37,
// and recomputing the mask from the frame with the forward's FMA chain 56 - as much as reading `out`; hence the words.
template <bool BITS>
__global__ __launch_bounds__(256) void k_enc_stem_wrw_mfma(const float *__restrict__ x, const float *__restrict__ out,
                                                           const uint32_t *__restrict__ relu_bits, const float *__restrict__ g_out,
                                                           float *__restrict__ partial,
                                                           float *__restrict__ partial_b, int64_t N) {
  __shared__ float tile[34 * ES_TS];
  // channel planes 260 floats apart: the masked gradient goes in as one ds_write_b128 per 16 bytes loaded.  With the 257 of the
  // VALU version a wave's 64 scalar writes hit 8 banks - 32 eight-way-conflicting writes per thread and frame were the whole
  // kernel: 158 us whether the products ran on the VALU or the matrix cores, with or without `out`, with or without prefetch.
  constexpr int GS = 260;
  __shared__ __attribute__((aligned(16))) float gt[ES_CO * GS];
  __shared__ __attribute__((aligned(16))) uint32_t mw[256];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int tap = j < 9 ? j : 8;
  const float *tb = tile + (8 * wv) * ES_TS + 2 * g + (tap / 3) * ES_TS + tap % 3;   // pixel (4 wv + (s >> 2), 4 (s & 3) + g), tap j
  const float *ga = gt + j * GS + 64 * wv + g;                                     // channel j (+16), pixel 64 wv + 4 s + g
  es_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  // g_out / out of the NEXT frame are requested before this frame's MFMAs (16 x 16 bytes per thread in flight): fetched in a
  // loop that also writes LDS, hipcc waits for every pair of loads before the next pair is issued - eight memory round trips
  // per frame, which is what bound the VALU version (158 us with either arithmetic)
  float4 gv[8], ov[8], xv;
  uint32_t mv = 0;
  auto fetch = [&](int64_t n) {
    const bool ok = n < N;
    const float4 *g4 = reinterpret_cast<const float4 *>(g_out + (ok ? n : 0) * ES_CO * 256);
    const float4 *o4 = reinterpret_cast<const float4 *>(out + (ok ? n : 0) * ES_CO * 256);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      gv[q] = g4[threadIdx.x + 256 * q];
      if constexpr (!BITS) ov[q] = o4[threadIdx.x + 256 * q];
    }
    xv = reinterpret_cast<const float4 *>(x + (ok ? n : 0) * 1024)[threadIdx.x];
    if constexpr (BITS) mv = relu_bits[(ok ? n : 0) * 256 + threadIdx.x];
  };
  fetch(blockIdx.x);
  for (int i = threadIdx.x; i < 34 * ES_TS; i += 256) tile[i] = 0.f;   // the border stays zero for the whole kernel
  for (int64_t n = blockIdx.x; n < N; n += gridDim.x) {
    __syncthreads();
    {                                                 // frame row h = tid / 8, columns 4 (tid & 7) ..: interior of the zero-bordered tile
      float *d = tile + ((threadIdx.x >> 3) + 1) * ES_TS + 4 * (threadIdx.x & 7) + 1;
      d[0] = xv.x; d[1] = xv.y; d[2] = xv.z; d[3] = xv.w;
    }
    if constexpr (BITS) {
      mw[threadIdx.x] = mv;
      __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = (threadIdx.x + 256 * q) * 4, c = e >> 8, hw = e & 255;
      float4 md;
      if constexpr (BITS) {
        const uint4 m = *reinterpret_cast<const uint4 *>(mw + hw);
        md = make_float4((m.x >> c) & 1u ? gv[q].x : 0.f, (m.y >> c) & 1u ? gv[q].y : 0.f, (m.z >> c) & 1u ? gv[q].z : 0.f,
                         (m.w >> c) & 1u ? gv[q].w : 0.f);
      } else {
        md = make_float4(ov[q].x > 0.f ? gv[q].x : 0.f, ov[q].y > 0.f ? gv[q].y : 0.f, ov[q].z > 0.f ? gv[q].z : 0.f,
                         ov[q].w > 0.f ? gv[q].w : 0.f);
      }
      *reinterpret_cast<float4 *>(gt + c * GS + hw) = md;
    }
    __syncthreads();
    fetch(n + gridDim.x);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float xb = tb[2 * (s >> 2) * ES_TS + 8 * (s & 3)];
      const float b = j == 9 ? 1.0f : xb;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[4 * s], b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[16 * GS + 4 * s], b, acc1, 0, 0, 0);
    }
  }
  __syncthreads();
  float *red = gt;   // [wave][channel 32][column 16]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[(wv * 32 + 4 * g + r) * 16 + j] = acc0[r];
    red[(wv * 32 + 16 + 4 * g + r) * 16 + j] = acc1[r];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < ES_CO * 10; o += 256) {
    const int c = o / 10, k = o - c * 10;
    const float s = (red[c * 16 + k] + red[(32 + c) * 16 + k]) + (red[(64 + c) * 16 + k] + red[(96 + c) * 16 + k]);
    if (k < 9) partial[(int64_t)blockIdx.x * (ES_CO * 9) + c * 9 + k] = s;
    else partial_b[(int64_t)blockIdx.x * ES_CO + c] = s;
  }
}

}  // namespace kvae
