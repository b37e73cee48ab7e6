// kvae_vae.hip — the frame VAE's hand-written kernels and the optimizer step (the callers either side of the
// section-8 path): conv epilogues and column sums (vae_epilogue.h), the fused clip + Adam step, the Bernoulli frame term
// (vae_loss.h), the direct / f32-MFMA / Winograd convolutions (vae_conv_edge.h, vae_conv_mid.h, vae_conv_up_wino.h) and the heads
// (vae_heads.h), with their C-ABI entry points.  A unit of its own since round 3: half of what used to be one 1300-line unit,
// compiled side by side with the LGSSM units (which carry per-unit scheduler flags, __graft_entry__.UNIT_FLAGS; these
// kernels - hand-placed sched_barrier pipelines at full occupancy - keep the compiler's default strategy).
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/kvae_lgssm.h"

namespace kvae {}
using namespace kvae;

extern "C" int kvae_launch_status(const char *what);   // kvae_lgssm.hip: hipGetLastError -> KVAE_OK / KVAE_ERR_LAUNCH (+ kvae_last_error)
static int launch_status(const char *what) { return kvae_launch_status(what); }

// ---------------------------------------------------------------------------------------------
// fused conv epilogues of the frame VAE (vae_epilogue.h)
// ---------------------------------------------------------------------------------------------
#include "vae_epilogue.h"

// out[c] = sum_r partials[r, c]: second stage of every deterministic two-stage reduction (bias / weight gradient partial
// rows).  HBM-bound (the 32->128 layers hand over 256 x 36864 floats = 38 MB): a lane owns four columns (dwordx4, 1 KiB per
// wave and row), the eight waves of a block take every eighth row, four rows in flight per wave, fold through LDS.
__device__ __forceinline__ void colsum_v4_body(const float *__restrict__ partials, float *__restrict__ out, int64_t rows, int64_t cols,
                                               unsigned block) {
  __shared__ float4 red[8][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t c = ((int64_t)block * 64 + lane) * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (c < cols) {
    const float *p = partials + c;
    int64_t r = wv;
    for (; r + 24 < rows; r += 32) {
      const float4 a = *reinterpret_cast<const float4 *>(p + r * cols), b = *reinterpret_cast<const float4 *>(p + (r + 8) * cols),
                   d = *reinterpret_cast<const float4 *>(p + (r + 16) * cols), e = *reinterpret_cast<const float4 *>(p + (r + 24) * cols);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
      s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
      s2.x += d.x; s2.y += d.y; s2.z += d.z; s2.w += d.w;
      s3.x += e.x; s3.y += e.y; s3.z += e.z; s3.w += e.w;
    }
    for (; r < rows; r += 8) {
      const float4 a = *reinterpret_cast<const float4 *>(p + r * cols);
      s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    }
  }
  red[wv][lane] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z),
                              (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (wv == 0 && c < cols) {
    float4 t = red[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) { const float4 u = red[w][lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *reinterpret_cast<float4 *>(out + c) = t;
  }
}
__global__ __launch_bounds__(512) void k_colsum_v4(const float *__restrict__ partials, float *__restrict__ out, int64_t rows,
                                                   int64_t cols) {
  colsum_v4_body(partials, out, rows, cols, blockIdx.x);
}
// two jobs in one launch: the first nb_a workgroups take job a, the rest job b
__global__ __launch_bounds__(512) void k_colsum_v4_pair(const float *__restrict__ pa, float *__restrict__ oa, int64_t rows_a, int64_t cols_a,
                                                        unsigned nb_a, const float *__restrict__ pb, float *__restrict__ ob,
                                                        int64_t rows_b, int64_t cols_b) {
  if (blockIdx.x < nb_a) colsum_v4_body(pa, oa, rows_a, cols_a, blockIdx.x);
  else colsum_v4_body(pb, ob, rows_b, cols_b, blockIdx.x - nb_a);
}
// any column count (scalar): 64 columns x 4 row lanes per block
__global__ __launch_bounds__(256) void k_colsum(const float *__restrict__ partials, float *__restrict__ out, int64_t rows,
                                                int64_t cols) {
  __shared__ float red[256];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + cx;
  float s0 = 0.f, s1 = 0.f;
  if (c < cols) {
    int64_t r = ry;
    for (; r + 4 < rows; r += 8) {
      s0 += partials[r * cols + c];
      s1 += partials[(r + 4) * cols + c];
    }
    if (r < rows) s0 += partials[r * cols + c];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (ry == 0 && c < cols) out[c] = (red[cx] + red[64 + cx]) + (red[128 + cx] + red[192 + cx]);
}

// Forward, R = 1 or 2: each thread produces 4 consecutive outputs along W (one 16-byte store) from one 16-byte
// (R = 1) or two 8-byte (R = 2: the two sub-pixel channels dx = 0,1 of this output row) loads; 32-bit index math.
template <int R>
__global__ __launch_bounds__(256) void k_vae_epilogue_fwd_v4(const float *__restrict__ in, const float *__restrict__ bias,
                                                             float *__restrict__ out, int C, int H, int W, int64_t quads,
                                                             int relu) {
  const int OW = W * R, OH = H * R, QW = OW / 4;
  for (int64_t qd = (int64_t)blockIdx.x * 256 + threadIdx.x; qd < quads; qd += (int64_t)gridDim.x * 256) {
    const int qw = (int)(qd % QW);
    const int64_t t1 = qd / QW;
    const int oh = (int)(t1 % OH);
    const int64_t t2 = t1 / OH;
    const int c = (int)(t2 % C);
    const int64_t n = t2 / C;
    float4 v;
    if (R == 1) {
      const float b = bias[c];
      const float4 x = *reinterpret_cast<const float4 *>(in + ((n * C + c) * H + oh) * W + 4 * qw);
      v = make_float4(x.x + b, x.y + b, x.z + b, x.w + b);
    } else {
      const int ch0 = c * 4 + (oh & 1) * 2;
      const int64_t base = ((n * (C * 4) + ch0) * H + (oh >> 1)) * W + 2 * qw;
      const float2 a = *reinterpret_cast<const float2 *>(in + base);                    // dx = 0, w = 2qw, 2qw+1
      const float2 d = *reinterpret_cast<const float2 *>(in + base + (int64_t)H * W);  // dx = 1
      const float b0 = bias[ch0], b1 = bias[ch0 + 1];
      v = make_float4(a.x + b0, d.x + b1, a.y + b0, d.y + b1);
    }
    if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    *reinterpret_cast<float4 *>(out + qd * 4) = v;
  }
}

__global__ __launch_bounds__(256) void k_vae_epilogue_fwd(const float *in, const float *bias, float *out, EpiShape s,
                                                          int64_t total, int relu) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256)
    epi_fwd_elem(s, in, bias, out, o, relu);
}

// Backward with the bias gradient folded in: a block owns 256 consecutive output positions of the per-sample volume
// [C, H*r, W*r] and walks a chunk of samples; per-thread sums go to per-channel LDS bins (ds_add_f32), one partial row
// per (chunk, channel) leaves the block -> bias_partials[chunk, C*r*r] (summed by the caller; fixed order per bin is not
// guaranteed inside a block: float LDS atomics, differences are at rounding level).
__global__ __launch_bounds__(256) void k_vae_epilogue_bwd_bias(const float *__restrict__ g_out, const float *__restrict__ out,
                                                               float *__restrict__ g_in, float *__restrict__ bias_partials,
                                                               EpiShape s, int n_per_chunk, int relu) {
  extern __shared__ float bins[];   // C*r*r floats
  const int Cin = s.C * s.r * s.r;
  const int OW = s.W * s.r, OH = s.H * s.r;
  const int64_t vol = (int64_t)s.C * OH * OW;
  for (int i = threadIdx.x; i < Cin; i += 256) bins[i] = 0.f;
  __syncthreads();
  const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;   // position inside one sample's output volume
  if (pos < vol) {
    const int ow = (int)(pos % OW);
    const int oh = (int)((pos / OW) % OH);
    const int c = (int)(pos / ((int64_t)OW * OH));
    const int ch = c * s.r * s.r + (oh % s.r) * s.r + (ow % s.r);
    const int64_t in_off = ((int64_t)ch * s.H + oh / s.r) * s.W + ow / s.r;
    const int64_t in_vol = (int64_t)Cin * s.H * s.W;
    const int64_t n0 = (int64_t)blockIdx.y * n_per_chunk;
    const int64_t n1 = n0 + n_per_chunk < s.N ? n0 + n_per_chunk : s.N;
    float acc = 0.f;
    for (int64_t n = n0; n < n1; ++n) {
      float g = g_out[n * vol + pos];
      if (relu && !(out[n * vol + pos] > 0.f)) g = 0.f;
      g_in[n * in_vol + in_off] = g;
      acc += g;
    }
    atomicAdd(&bins[ch], acc);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cin; i += 256) {
    // only the channels this block touched are non-zero; every block writes its own disjoint partial row slice
    if (bins[i] != 0.f) atomicAdd(&bias_partials[(int64_t)blockIdx.y * Cin + i], bins[i]);
  }
}

__global__ __launch_bounds__(256) void k_vae_epilogue_bwd(const float *g_out, const float *out, float *g_in, EpiShape s,
                                                          int64_t total, int relu) {
  for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256)
    epi_bwd_elem(s, g_out, out, g_in, o, relu);
}

#define KVAE_EPI_SAMPLES_PER_CHUNK 32
extern "C" int64_t kvae_bias_partial_rows(int64_t N);
static unsigned epi_grid(int64_t total) {
  const int64_t blocks = (total + 255) / 256;
  return (unsigned)(blocks < 256 * 32 ? blocks : 256 * 32);   // <= 32 blocks per CU, grid-stride the rest
}

extern "C" {
int kvae_bias_shuffle_act_fwd(const float *in, const float *bias, float *out, int64_t N, int32_t C, int32_t H, int32_t W,
                              int32_t r, int32_t relu, void *stream) {
  if (!in || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || r < 1) return KVAE_ERR_ARG;
  const EpiShape s{N, C, H, W, r};
  const int64_t total = N * C * H * W * r * r;
  const bool aligned = ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (aligned && r == 1 && W % 4 == 0)
    k_vae_epilogue_fwd_v4<1><<<dim3(epi_grid(total / 4)), dim3(256), 0, (hipStream_t)stream>>>(in, bias, out, C, H, W, total / 4, relu);
  else if (aligned && r == 2 && W % 2 == 0)
    k_vae_epilogue_fwd_v4<2><<<dim3(epi_grid(total / 4)), dim3(256), 0, (hipStream_t)stream>>>(in, bias, out, C, H, W, total / 4, relu);
  else
    k_vae_epilogue_fwd<<<dim3(epi_grid(total)), dim3(256), 0, (hipStream_t)stream>>>(in, bias, out, s, total, relu);
  return launch_status("k_vae_epilogue_fwd");
}
int kvae_bias_shuffle_act_bwd(const float *g_out, const float *out, float *g_in, float *bias_partials, int64_t N, int32_t C,
                              int32_t H, int32_t W, int32_t r, int32_t relu, void *stream) {
  if (!g_out || !g_in || (relu && !out)) return KVAE_ERR_NULL;
  if (N < 1 || C < 1 || H < 1 || W < 1 || r < 1) return KVAE_ERR_ARG;
  const EpiShape s{N, C, H, W, r};
  const int64_t total = N * C * H * W * r * r;
  hipStream_t st = (hipStream_t)stream;
  if (bias_partials) {
    const int Cin = C * r * r;
    const int64_t chunks = kvae_bias_partial_rows(N);
    if (hipMemsetAsync(bias_partials, 0, sizeof(float) * chunks * Cin, st) != hipSuccess) return launch_status("memset bias_partials");
    const int64_t vol = (int64_t)C * H * W * r * r;
    k_vae_epilogue_bwd_bias<<<dim3((unsigned)((vol + 255) / 256), (unsigned)chunks), dim3(256), sizeof(float) * Cin, st>>>(
        g_out, out, g_in, bias_partials, s, KVAE_EPI_SAMPLES_PER_CHUNK, relu);
    return launch_status("k_vae_epilogue_bwd_bias");
  }
  k_vae_epilogue_bwd<<<dim3(epi_grid(total)), dim3(256), 0, st>>>(g_out, out, g_in, s, total, relu);
  return launch_status("k_vae_epilogue_bwd");
}
int kvae_colsum(const float *partials, float *out, int64_t rows, int64_t cols, void *stream) {
  if (!partials || !out) return KVAE_ERR_NULL;
  if (rows < 1 || cols < 1) return KVAE_ERR_ARG;
  if ((cols & 3) == 0 && ((((uintptr_t)partials | (uintptr_t)out) & 15) == 0))
    k_colsum_v4<<<dim3((unsigned)((cols + 255) / 256)), dim3(512), 0, (hipStream_t)stream>>>(partials, out, rows, cols);
  else
    k_colsum<<<dim3((unsigned)((cols + 63) / 64)), dim3(256), 0, (hipStream_t)stream>>>(partials, out, rows, cols);
  return launch_status("k_colsum");
}
int kvae_colsum2(const float *pa, float *oa, int64_t rows_a, int64_t cols_a, const float *pb, float *ob, int64_t rows_b,
                 int64_t cols_b, void *stream) {
  if (!pa || !oa || !pb || !ob) return KVAE_ERR_NULL;
  if (rows_a < 1 || cols_a < 1 || rows_b < 1 || cols_b < 1) return KVAE_ERR_ARG;
  if (((cols_a | cols_b) & 3) == 0 && ((((uintptr_t)pa | (uintptr_t)oa | (uintptr_t)pb | (uintptr_t)ob) & 15) == 0)) {
    const unsigned nb_a = (unsigned)((cols_a + 255) / 256), nb_b = (unsigned)((cols_b + 255) / 256);
    k_colsum_v4_pair<<<dim3(nb_a + nb_b), dim3(512), 0, (hipStream_t)stream>>>(pa, oa, rows_a, cols_a, nb_a, pb, ob, rows_b, cols_b);
    return launch_status("k_colsum_v4_pair");
  }
  const int rc = kvae_colsum(pa, oa, rows_a, cols_a, stream);
  return rc ? rc : kvae_colsum(pb, ob, rows_b, cols_b, stream);
}
int64_t kvae_bias_partial_rows(int64_t N) { return (N + KVAE_EPI_SAMPLES_PER_CHUNK - 1) / KVAE_EPI_SAMPLES_PER_CHUNK; }
}  // extern "C"

// ---------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam on flat buffers: two launches instead of ~12 (norm, clamp, reciprocal, scale, three foreach kernels)
// ---------------------------------------------------------------------------------------------
constexpr int CA_BLOCKS = 512;   // partial sums of squares (ws[0..CA_BLOCKS))
constexpr int CA_MAX_SEG = 1024;  // parameter tensors ("segments") of one flat buffer
// Segments: seg_of[i] names the parameter tensor element i belongs to; a segment with seg_active[s] == 0 is a FROZEN parameter
// (requires_grad False: the reference's training phases, train.py:142-207) - torch's clip_grad_norm_ and Adam skip it because its
// .grad is None: it adds nothing to the norm, its moments and its step count stay as they are.  Every segment counts its own steps
// (torch keeps `step` per parameter, so a parameter thawed at epoch 6 starts its bias correction at step 1).
__global__ __launch_bounds__(256) void k_grad_sumsq(const float *__restrict__ g, int64_t n, const int32_t *__restrict__ seg_of,
                                                    int n_seg, const float *__restrict__ seg_active,
                                                    const float *__restrict__ div_dev, float *__restrict__ seg_steps,
                                                    float *__restrict__ ws) {
  __shared__ float red[256];
  const float inv = div_dev ? 1.0f / fmaxf(*div_dev, 1.0f) : 1.0f;
  const bool gated = seg_of && seg_active;
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float v = g[i] * inv;
    if (gated && seg_active[seg_of[i]] == 0.f) v = 0.f;
    s = fmaf(v, v, s);
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) ws[blockIdx.x] = red[0];
  if (blockIdx.x == 0)                                  // the next launch reads the incremented step counts
    for (int sg = threadIdx.x; sg < n_seg; sg += 256)
      if (!seg_active || seg_active[sg] != 0.f) seg_steps[sg] += 1.0f;
}
__global__ __launch_bounds__(256) void k_clip_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                   float *__restrict__ v, int64_t n, const int32_t *__restrict__ seg_of, int n_seg,
                                                   const float *__restrict__ seg_active, const float *__restrict__ seg_steps,
                                                   const float *__restrict__ lr_dev, float lr, float beta1, float beta2, float eps,
                                                   float wd, float clip, const float *__restrict__ div_dev,
                                                   float *__restrict__ norm_out, const float *__restrict__ ws, int nparts) {
  __shared__ float red[256];
  __shared__ float s_step_size[CA_MAX_SEG], s_bc2s[CA_MAX_SEG];   // step_size < 0 marks a frozen segment
  float s = 0.f;                                      // every block folds the same partials in the same order
  for (int i = threadIdx.x; i < nparts; i += 256) s += ws[i];
  red[threadIdx.x] = s;
  const float lrv = lr_dev ? *lr_dev : lr;
  for (int sg = threadIdx.x; sg < n_seg; sg += 256) {
    const float step = seg_steps[sg];
    const bool on = !seg_active || seg_active[sg] != 0.f;
    s_step_size[sg] = on ? lrv / (1.0f - powf(beta1, step)) : -1.0f;
    s_bc2s[sg] = on ? sqrtf(1.0f - powf(beta2, step)) : 1.0f;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float total = sqrtf(red[0]);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = total;
  const float inv = div_dev ? 1.0f / fmaxf(*div_dev, 1.0f) : 1.0f;
  const float scale = inv * (clip > 0.f ? fminf(clip / (total + 1e-6f), 1.0f) : 1.0f);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int sg = seg_of ? seg_of[i] : 0;
    const float step_size = s_step_size[sg], bc2s = s_bc2s[sg];
    if (step_size < 0.f) continue;
    float gi = g[i] * scale;
    const float pi = p[i];
    if (wd != 0.f) gi = fmaf(wd, pi, gi);
    const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);          // lerp
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    m[i] = mi, v[i] = vi;
    p[i] = pi - step_size * mi / (sqrtf(vi) / bc2s + eps);
  }
}
extern "C" int kvae_clip_adam(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                              const int32_t *seg_of, int32_t n_seg, const float *seg_active, float *seg_steps, const float *lr_dev,
                              float lr, float beta1, float beta2, float eps, float weight_decay, float clip, const float *div_dev,
                              float *norm_out, float *ws, void *stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !seg_steps || !ws) return KVAE_ERR_NULL;
  if (n < 1 || n_seg < 1 || n_seg > CA_MAX_SEG || (!seg_of && n_seg != 1)) return KVAE_ERR_ARG;
  const int64_t want = (n + 1023) / 1024;
  const unsigned parts = (unsigned)(want < CA_BLOCKS ? want : CA_BLOCKS);
  k_grad_sumsq<<<dim3(parts), dim3(256), 0, (hipStream_t)stream>>>(grads, n, seg_of, n_seg, seg_active, div_dev, seg_steps, ws);
  int rc = launch_status("k_grad_sumsq");
  if (rc) return rc;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  k_clip_adam<<<dim3(blocks), dim3(256), 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, seg_of, n_seg, seg_active,
                                                                   seg_steps, lr_dev, lr, beta1, beta2, eps, weight_decay, clip,
                                                                   div_dev, norm_out, ws, (int)parts);
  return launch_status("k_clip_adam");
}

// ---------------------------------------------------------------------------------------------
// fused Bernoulli reconstruction term (vae_loss.h): one wavefront per frame
// ---------------------------------------------------------------------------------------------
#include "vae_loss.h"

__global__ __launch_bounds__(256) void k_vae_bce_fwd(const float *__restrict__ logits, const float *__restrict__ x,
                                                     float *__restrict__ frame_ll, int64_t frames, int pixels) {
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (f >= frames) return;
  const float *l = logits + f * pixels, *t = x + f * pixels;
  float acc = 0.f;
  if ((pixels & 3) == 0 && (((uintptr_t)l | (uintptr_t)t) & 15) == 0) {
    for (int i = lane * 4; i < pixels; i += 256) {
      const float4 a = *reinterpret_cast<const float4 *>(l + i), b = *reinterpret_cast<const float4 *>(t + i);
      acc += bce_logit(a.x, b.x) + bce_logit(a.y, b.y) + bce_logit(a.z, b.z) + bce_logit(a.w, b.w);
    }
  } else {
    for (int i = lane; i < pixels; i += 64) acc += bce_logit(l[i], t[i]);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) frame_ll[f] = -acc;
}

__global__ __launch_bounds__(256) void k_vae_bce_bwd(const float *__restrict__ logits, const float *__restrict__ x,
                                                     const float *__restrict__ g_frame, float *__restrict__ g_logits,
                                                     int64_t total, int pixels) {
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (int64_t)gridDim.x * 1024) {
    const float g = -g_frame[i / pixels];   // pixels % 4 == 0 on this path: the 4 elements share a frame
    const float4 a = *reinterpret_cast<const float4 *>(logits + i), b = *reinterpret_cast<const float4 *>(x + i);
    *reinterpret_cast<float4 *>(g_logits + i) = make_float4(g * (sigmoid_stable(a.x) - b.x), g * (sigmoid_stable(a.y) - b.y),
                                                            g * (sigmoid_stable(a.z) - b.z), g * (sigmoid_stable(a.w) - b.w));
  }
}
__global__ __launch_bounds__(256) void k_vae_bce_bwd_scalar(const float *logits, const float *x, const float *g_frame,
                                                            float *g_logits, int64_t total, int pixels) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    g_logits[i] = -g_frame[i / pixels] * (sigmoid_stable(logits[i]) - x[i]);
}

extern "C" {
int kvae_bce_frames_fwd(const float *logits, const float *x, float *frame_ll, int64_t frames, int32_t pixels, void *stream) {
  if (!logits || !x || !frame_ll) return KVAE_ERR_NULL;
  if (frames < 1 || pixels < 1) return KVAE_ERR_ARG;
  k_vae_bce_fwd<<<dim3((unsigned)((frames + 3) / 4)), dim3(256), 0, (hipStream_t)stream>>>(logits, x, frame_ll, frames, pixels);
  return launch_status("k_vae_bce_fwd");
}
int kvae_bce_frames_bwd(const float *logits, const float *x, const float *g_frame, float *g_logits, int64_t frames,
                        int32_t pixels, void *stream) {
  if (!logits || !x || !g_frame || !g_logits) return KVAE_ERR_NULL;
  if (frames < 1 || pixels < 1) return KVAE_ERR_ARG;
  const int64_t total = frames * pixels;
  const bool v4 = (pixels & 3) == 0 && ((((uintptr_t)logits | (uintptr_t)x | (uintptr_t)g_logits) & 15) == 0);
  if (v4)
    k_vae_bce_bwd<<<dim3(epi_grid(total / 4)), dim3(256), 0, (hipStream_t)stream>>>(logits, x, g_frame, g_logits, total, pixels);
  else
    k_vae_bce_bwd_scalar<<<dim3(epi_grid(total)), dim3(256), 0, (hipStream_t)stream>>>(logits, x, g_frame, g_logits, total, pixels);
  return launch_status("k_vae_bce_bwd");
}
}  // extern "C"

#include "vae_conv_edge.h"
extern "C" {
int64_t kvae_conv_edge_partial_rows(int64_t N) { return N < 1024 ? (N < 1 ? 1 : N) : 1024; }   // four workgroups per CU

int kvae_dec_head_fwd(const float *in, const float *W, const float *bias, float *logits, float *w_scratch, int64_t N,
                      int32_t Cin, int32_t side, void *stream) {
  if (!in || !W || !bias || !logits || !w_scratch) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != DH_CI || side != DH_S) return KVAE_ERR_DIMS;
  k_dec_head_prep<<<dim3(1), dim3(256), 0, (hipStream_t)stream>>>(W, w_scratch);
  k_dec_head_fwd<<<dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream>>>(in, w_scratch, bias, logits);
  return launch_status("k_dec_head_fwd");
}
int kvae_dec_head_bwd(const float *in, const float *W, const float *g_logits, float *g_in, float *w_partials,
                      float *b_partials, float *w_scratch, int64_t N, int32_t Cin, int32_t side, void *stream) {
  if (!in || !W || !g_logits || !w_partials || !b_partials || !w_scratch) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cin != DH_CI || side != DH_S) return KVAE_ERR_DIMS;
  if (g_in) {
    k_dec_head_prep<<<dim3(1), dim3(256), 0, (hipStream_t)stream>>>(W, w_scratch);
    k_dec_head_bwd_data<<<dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream>>>(g_logits, w_scratch, g_in);
    const int rc = launch_status("k_dec_head_bwd_data");
    if (rc) return rc;
  }
  k_dec_head_wrw<<<dim3((unsigned)kvae_conv_edge_partial_rows(N)), dim3(256), 0, (hipStream_t)stream>>>(in, g_logits, w_partials,
                                                                                                     b_partials, N);
  return launch_status("k_dec_head_wrw");
}
int kvae_enc_stem_fwd(const float *x, const float *W, const float *bias, float *out, uint32_t *relu_bits, int64_t N, int32_t Cout,
                      int32_t side, void *stream) {
  if (!x || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cout != ES_CO || side != ES_IN) return KVAE_ERR_DIMS;
  k_enc_stem_fwd<<<dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream>>>(x, W, bias, out, relu_bits);
  return launch_status("k_enc_stem_fwd");
}
int kvae_enc_stem_bwd(const float *x, const float *out, const uint32_t *relu_bits, const float *g_out, float *w_partials,
                      float *b_partials, int64_t N, int32_t Cout, int32_t side, void *stream) {
  if (!x || (!out && !relu_bits) || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (Cout != ES_CO || side != ES_IN) return KVAE_ERR_DIMS;
  static const int mfma = getenv("KVAE_STEM_MFMA") ? atoi(getenv("KVAE_STEM_MFMA")) : 1;   // 0: VALU version, 2: mask from out (A/B runs)
  const dim3 grid((unsigned)kvae_conv_edge_partial_rows(N));
  if (mfma == 1 && relu_bits) k_enc_stem_wrw_mfma<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, relu_bits, g_out, w_partials, b_partials, N);
  else if (!out) return KVAE_ERR_NULL;
  else if (mfma) k_enc_stem_wrw_mfma<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, relu_bits, g_out, w_partials, b_partials, N);
  else k_enc_stem_wrw<<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
  return launch_status("k_enc_stem_wrw");
}
}  // extern "C"

#include "vae_conv_mid.h"
static inline int64_t enc_mid_grid(int64_t N, int32_t side) {
  const int64_t fpi = side == 16 ? 2 : 8, iters = (N + fpi - 1) / fpi;
  return iters < 256 ? (iters < 1 ? 1 : iters) : 256;   // one persistent workgroup per CU
}
extern "C" {
int64_t kvae_enc_mid_partial_rows(int64_t N, int32_t side) { return enc_mid_grid(N, side); }

int kvae_enc_mid_fwd(const float *in, const float *W, const float *bias, float *out, int64_t N, int32_t C, int32_t side,
                     void *stream) {
  if (!in || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * EM_C * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;   // 32-bit byte offsets
  if (C != EM_C || (side != 16 && side != 8)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)enc_mid_grid(N, side));
  if (side == 16) k_enc_mid_fwd<16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, W, bias, out, N);
  else k_enc_mid_fwd<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, W, bias, out, N);
  return launch_status("k_enc_mid_fwd");
}
int kvae_enc_mid_bwd(const float *in, const float *W, const float *out, const float *g_out, float *g_in,
                     float *w_partials, float *b_partials, int64_t N, int32_t C, int32_t side, void *stream) {
  if (!in || !W || !out || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * EM_C * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;
  if (C != EM_C || (side != 16 && side != 8)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)enc_mid_grid(N, side));
  if (g_in) {
    if (side == 16) k_enc_mid_bwd_data<16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_in, N);
    else k_enc_mid_bwd_data<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_in, N);
    const int rc = launch_status("k_enc_mid_bwd_data");
    if (rc) return rc;
  }
  if (side == 16) k_enc_mid_wrw<16><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, out, g_out, w_partials, b_partials, N);
  else k_enc_mid_wrw<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(in, out, g_out, w_partials, b_partials, N);
  return launch_status("k_enc_mid_wrw");
}
}  // extern "C"

#include "vae_conv_up_wino.h"
static bool dec_up_wino() {
  static const int env = getenv("KVAE_WINO") ? atoi(getenv("KVAE_WINO")) : 1;   // 0: direct convolution (A/B runs)
  return env != 0;
}
// Persistent decoder-block workgroups: one per CU, or fewer (KVAE_UP_WGS) so that a second stream's kernels that do not fit
// beside them (anything above 32 registers per lane) find free CUs while they run.
static int g_dec_up_wgs = 0;   // 0: not set through kvae_dec_up_set_workgroups
static inline int64_t dec_up_cap() {
  static const int env = getenv("KVAE_UP_WGS") ? atoi(getenv("KVAE_UP_WGS")) : 0;   // the environment wins over the setter (A/B runs)
  const int v = env >= 1 && env <= 256 ? env : __atomic_load_n(&g_dec_up_wgs, __ATOMIC_RELAXED);
  return v >= 1 && v <= 256 ? v : 256;
}
static inline int64_t dec_up_grid(int64_t N, int32_t side) {
  const int64_t fpi = side == 8 ? 2 : 8, iters = (N + fpi - 1) / fpi, cap = dec_up_cap();
  return iters < cap ? (iters < 1 ? 1 : iters) : cap;
}
extern "C" {
int64_t kvae_dec_up_partial_rows(int64_t N, int32_t side) { return dec_up_grid(N, side); }
int32_t kvae_dec_up_set_workgroups(int32_t n) {
  const int32_t prev = (int32_t)dec_up_cap();
  __atomic_store_n(&g_dec_up_wgs, n >= 1 && n <= 256 ? n : 0, __ATOMIC_RELAXED);
  return prev;
}

int kvae_dec_up_fwd(const float *x, const float *W, const float *bias, float *out, int64_t N, int32_t Cin, int32_t side,
                    void *stream) {
  if (!x || !W || !bias || !out) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * UP_CO * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;   // 32-bit byte offsets
  if (Cin != UP_CI || (side != 8 && side != 4)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)dec_up_grid(N, side));
  if (dec_up_wino()) {   // pairs of workgroups (one per half of the output channels) walk the column sets together
    const int64_t sets = side == 8 ? N : (N + 3) / 4;
    const dim3 wgrid((unsigned)(sets < dec_up_cap() ? sets : dec_up_cap()));
    if (side == 8) k_dec_up_fwd_wino<8><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
    else k_dec_up_fwd_wino<4><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
    return launch_status("k_dec_up_fwd_wino");
  }
  if (side == 8) k_dec_up_fwd<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
  else k_dec_up_fwd<4><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, W, bias, out, N);
  return launch_status("k_dec_up_fwd");
}
int kvae_dec_up_bwd(const float *x, const float *W, const float *out, const float *g_out, float *g_x, float *w_partials,
                    float *b_partials, int64_t N, int32_t Cin, int32_t side, void *stream) {
  if (!x || !W || !out || !g_out || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1 || (N + 256 * 8) * UP_CO * side * side * 4 >= EM_MAX_BYTES) return KVAE_ERR_ARG;
  if (Cin != UP_CI || (side != 8 && side != 4)) return KVAE_ERR_DIMS;
  const dim3 grid((unsigned)dec_up_grid(N, side));
  if (g_x) {
    const int64_t sets = side == 8 ? N : (N + 3) / 4;
    const dim3 wgrid((unsigned)(sets < dec_up_cap() ? sets : dec_up_cap()));
    if (dec_up_wino() && side == 8) k_dec_up_bwd_data_wino<8><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    else if (dec_up_wino()) k_dec_up_bwd_data_wino<4><<<wgrid, dim3(512), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    else if (side == 8) k_dec_up_bwd_data<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    else k_dec_up_bwd_data<4><<<grid, dim3(256), 0, (hipStream_t)stream>>>(W, out, g_out, g_x, N);
    const int rc = launch_status("k_dec_up_bwd_data");
    if (rc) return rc;
  }
  if (dec_up_wino()) {
    if (side == 8) k_dec_up_wrw_wino<8><<<grid, dim3(512), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
    else k_dec_up_wrw_wino<4><<<grid, dim3(512), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
    return launch_status("k_dec_up_wrw_wino");
  }
  if (side == 8) k_dec_up_wrw<8><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
  else k_dec_up_wrw<4><<<grid, dim3(256), 0, (hipStream_t)stream>>>(x, out, g_out, w_partials, b_partials, N);
  return launch_status("k_dec_up_wrw");
}
}  // extern "C"

#include "vae_heads.h"
extern "C" {
int64_t kvae_head_partial_rows(void) { return HD_WAVES; }

int kvae_enc_head_fwd(const float *feat, const float *Wmu, const float *bmu, const float *Wvar, const float *bvar,
                      const float *eps, float *mu, float *var, float *a, int64_t N, int32_t F, int32_t A,
                      float noise_emission, void *stream) {
  if (!feat || !Wmu || !bmu || !Wvar || !bvar || !mu || !var || !a) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  const unsigned grid = (unsigned)(N < 4 * 512 ? (N + 3) / 4 : 512);
  k_enc_head_fwd<<<dim3(grid), dim3(256), 0, (hipStream_t)stream>>>(feat, Wmu, bmu, Wvar, bvar, eps, noise_emission, mu, var, a, N);
  return launch_status("k_enc_head_fwd");
}
int kvae_enc_head_bwd(const float *feat, const float *Wmu, const float *Wvar, const float *var, const float *eps,
                      const float *g_a, const float *g_mu, const float *g_var, float *g_feat, float *w_partials,
                      float *b_partials, int64_t N, int32_t F, int32_t A, float noise_emission, void *stream) {
  if (!feat || !Wmu || !Wvar || !var || !g_feat || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  k_enc_head_bwd<<<dim3(HD_WAVES / 4), dim3(256), 0, (hipStream_t)stream>>>(feat, Wmu, Wvar, var, eps, g_a, g_mu, g_var,
                                                                             noise_emission, g_feat, w_partials, b_partials, N);
  return launch_status("k_enc_head_bwd");
}
int kvae_dec_fc_fwd(const float *a, const float *W, const float *b, float *h, int64_t N, int32_t F, int32_t A, void *stream) {
  if (!a || !W || !b || !h) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  k_dec_fc_fwd<<<dim3(epi_grid(N * (HD_F / 4))), dim3(256), 0, (hipStream_t)stream>>>(a, W, b, h, N * (HD_F / 4));
  return launch_status("k_dec_fc_fwd");
}
int kvae_dec_fc_bwd(const float *g_h, const float *a, const float *W, float *g_a, float *w_partials, float *b_partials,
                    int64_t N, int32_t F, int32_t A, void *stream) {
  if (!g_h || !a || !W || !g_a || !w_partials || !b_partials) return KVAE_ERR_NULL;
  if (N < 1) return KVAE_ERR_ARG;
  if (F != HD_F || A != HD_A) return KVAE_ERR_DIMS;
  k_dec_fc_bwd<<<dim3(HD_WAVES / 4), dim3(256), 0, (hipStream_t)stream>>>(g_h, a, W, g_a, w_partials, b_partials, N);
  return launch_status("k_dec_fc_bwd");
}
int kvae_latent_reg_fwd(const float *a, const float *mu, const float *var, float *reg, int64_t N, int32_t A, void *stream) {
  if (!a || !mu || !var || !reg) return KVAE_ERR_NULL;
  if (N < 1 || A < 1) return KVAE_ERR_ARG;
  k_latent_reg_fwd<<<dim3(epi_grid(N)), dim3(256), 0, (hipStream_t)stream>>>(a, mu, var, reg, N, A);
  return launch_status("k_latent_reg_fwd");
}
int kvae_latent_reg_bwd(const float *a, const float *mu, const float *var, const float *g, float *g_a, float *g_mu,
                        float *g_var, int64_t N, int32_t A, void *stream) {
  if (!a || !mu || !var || !g || !g_a || !g_mu || !g_var) return KVAE_ERR_NULL;
  if (N < 1 || A < 1) return KVAE_ERR_ARG;
  k_latent_reg_bwd<<<dim3(epi_grid(N * A)), dim3(256), 0, (hipStream_t)stream>>>(a, mu, var, g, g_a, g_mu, g_var, N, A);
  return launch_status("k_latent_reg_bwd");
}
int kvae_loss_head_fwd(const float *lpx, const float *regf, const float *mask, const float *elbo_kf, const float *beta,
                       float scale_reconstruction, float vae_weight, float kf_weight, const float *weights_dev, float *out6,
                       float *coef3, int64_t n, void *stream) {
  if (!lpx || !regf || !elbo_kf || !beta || !out6 || !coef3) return KVAE_ERR_NULL;
  if (n < 1) return KVAE_ERR_ARG;
  k_loss_head_fwd<<<dim3(1), dim3(1024), 0, (hipStream_t)stream>>>(lpx, regf, mask, elbo_kf, beta, scale_reconstruction, vae_weight,
                                                                  kf_weight, weights_dev, out6, coef3, n);
  return launch_status("k_loss_head_fwd");
}
int kvae_loss_head_bwd(const float *g_loss, const float *coef3, const float *mask, float *g_lpx, float *g_regf,
                       float *g_elbo_kf, int64_t n, void *stream) {
  if (!g_loss || !coef3 || !g_lpx || !g_regf || !g_elbo_kf) return KVAE_ERR_NULL;
  if (n < 1) return KVAE_ERR_ARG;
  k_loss_head_bwd<<<dim3(epi_grid(n)), dim3(256), 0, (hipStream_t)stream>>>(g_loss, coef3, mask, g_lpx, g_regf,
                                                                            g_elbo_kf, n);
  return launch_status("k_loss_head_bwd");
}
}  // extern "C"
