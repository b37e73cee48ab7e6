// vae_heads.h — the skinny fully-connected ends of the frame VAE and the latent regulariser, one kernel per direction
// instead of ~12 library GEMM launches (N = 12800 rows against 2 or 4 columns: 30-40 us each on a GEMM kernel) and ~40
// element-wise launches per training step.  Reference: kvae/vae/vae.py:33-41 (fc_mu, fc_var + Sigmoid, noise_emission),
// kvae/model/model.py:81-84 (reparameterisation, std = sqrt(var + 1e-6)), vae.py:88-90 (decoder fc),
// kvae/vae/losses.py:64-66 (log p(a) - log q(a|x)).
// Shapes are the reference's defaults: F = 512 features (32 x 4 x 4), A = 2 latent dimensions; one wave per frame row,
// a lane owns 8 of the 512 feature columns (two dwordx4), the weights it needs live in registers for the whole kernel,
// weight / bias gradients are accumulated per lane over the wave's rows and leave as one partial row per wave.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {

constexpr int HD_F = 512, HD_A = 2, HD_WAVES = 512;   // partial rows = waves of the backward grids

__device__ __forceinline__ float hd_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float hd_dot(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
__device__ __forceinline__ float4 hd_fma(float s, const float4 a, const float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

// mu = feat Wmu^T + bmu; var = ne * sigmoid(feat Wvar^T + bvar); a = mu + eps * sqrt(var + 1e-6) (eps may be NULL: a = mu)
__global__ __launch_bounds__(256) void k_enc_head_fwd(const float *__restrict__ feat, const float *__restrict__ Wmu,
                                                      const float *__restrict__ bmu, const float *__restrict__ Wvar,
                                                      const float *__restrict__ bvar, const float *__restrict__ eps, float ne,
                                                      float *__restrict__ mu, float *__restrict__ var, float *__restrict__ a,
                                                      int64_t N) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  float4 wm[HD_A][2], wv[HD_A][2];
#pragma unroll
  for (int j = 0; j < HD_A; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      wm[j][h] = reinterpret_cast<const float4 *>(Wmu + j * HD_F)[h * 64 + lane];
      wv[j][h] = reinterpret_cast<const float4 *>(Wvar + j * HD_F)[h * 64 + lane];
    }
  for (int64_t n = gw; n < N; n += nw) {
    const float4 f0 = reinterpret_cast<const float4 *>(feat + n * HD_F)[lane], f1 = reinterpret_cast<const float4 *>(feat + n * HD_F)[64 + lane];
#pragma unroll
    for (int j = 0; j < HD_A; ++j) {
      const float dm = hd_wave_sum(hd_dot(f0, wm[j][0]) + hd_dot(f1, wm[j][1]));
      const float dv = hd_wave_sum(hd_dot(f0, wv[j][0]) + hd_dot(f1, wv[j][1]));
      if (lane == 0) {
        const float m = dm + bmu[j], s = ne * (1.f / (1.f + expf(-(dv + bvar[j]))));
        mu[n * HD_A + j] = m;
        var[n * HD_A + j] = s;
        a[n * HD_A + j] = eps ? fmaf(eps[n * HD_A + j], sqrtf(s + 1e-6f), m) : m;
      }
    }
  }
}

// upstream g_a, g_mu, g_var (each may be NULL = 0) -> g_feat and per-wave partial rows of the weight / bias gradients:
// w_partials[wave, (0:mu|1:var, j, 512)], b_partials[wave, (mu_0, mu_1, var_0, var_1)]
__global__ __launch_bounds__(256) void k_enc_head_bwd(const float *__restrict__ feat, const float *__restrict__ Wmu,
                                                      const float *__restrict__ Wvar, const float *__restrict__ var,
                                                      const float *__restrict__ eps, const float *__restrict__ g_a,
                                                      const float *__restrict__ g_mu, const float *__restrict__ g_var, float ne,
                                                      float *__restrict__ g_feat, float *__restrict__ w_partials,
                                                      float *__restrict__ b_partials, int64_t N) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  float4 wm[HD_A][2], wv[HD_A][2], am[HD_A][2], av[HD_A][2];
  float ab[2 * HD_A];
#pragma unroll
  for (int j = 0; j < HD_A; ++j) {
    ab[j] = ab[HD_A + j] = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      wm[j][h] = reinterpret_cast<const float4 *>(Wmu + j * HD_F)[h * 64 + lane];
      wv[j][h] = reinterpret_cast<const float4 *>(Wvar + j * HD_F)[h * 64 + lane];
      am[j][h] = av[j][h] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int64_t n = gw; n < N; n += nw) {
    const float4 f0 = reinterpret_cast<const float4 *>(feat + n * HD_F)[lane], f1 = reinterpret_cast<const float4 *>(feat + n * HD_F)[64 + lane];
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
#pragma unroll
    for (int j = 0; j < HD_A; ++j) {
      const float ga = g_a ? g_a[n * HD_A + j] : 0.f, s = var[n * HD_A + j];
      const float gm = (g_mu ? g_mu[n * HD_A + j] : 0.f) + ga;
      const float gv = (g_var ? g_var[n * HD_A + j] : 0.f) + (eps ? ga * eps[n * HD_A + j] * 0.5f / sqrtf(s + 1e-6f) : 0.f);
      const float gs = gv * s * (1.f - s / ne);            // through ne * sigmoid
      o0 = hd_fma(gm, wm[j][0], hd_fma(gs, wv[j][0], o0));
      o1 = hd_fma(gm, wm[j][1], hd_fma(gs, wv[j][1], o1));
      am[j][0] = hd_fma(gm, f0, am[j][0]); am[j][1] = hd_fma(gm, f1, am[j][1]);
      av[j][0] = hd_fma(gs, f0, av[j][0]); av[j][1] = hd_fma(gs, f1, av[j][1]);
      ab[j] += gm;
      ab[HD_A + j] += gs;
    }
    reinterpret_cast<float4 *>(g_feat + n * HD_F)[lane] = o0;
    reinterpret_cast<float4 *>(g_feat + n * HD_F)[64 + lane] = o1;
  }
  float4 *wp = reinterpret_cast<float4 *>(w_partials + gw * (2 * HD_A * HD_F));
#pragma unroll
  for (int j = 0; j < HD_A; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      wp[(j * HD_F) / 4 + h * 64 + lane] = am[j][h];
      wp[((HD_A + j) * HD_F) / 4 + h * 64 + lane] = av[j][h];
    }
  if (lane < 2 * HD_A) b_partials[gw * (2 * HD_A) + lane] = ab[lane];   // wave-uniform values
}

// h[n,k] = sum_j a[n,j] W[k,j] + b[k]   (decoder fc: 2 -> 512)
__global__ __launch_bounds__(256) void k_dec_fc_fwd(const float *__restrict__ a, const float *__restrict__ W,
                                                    const float *__restrict__ b, float *__restrict__ h, int64_t total4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i >> 7;
    const int k4 = (int)(i & 127);
    const float a0 = a[n * HD_A], a1 = a[n * HD_A + 1];
    const float4 w01 = reinterpret_cast<const float4 *>(W)[2 * k4], w23 = reinterpret_cast<const float4 *>(W)[2 * k4 + 1];
    const float4 bb = reinterpret_cast<const float4 *>(b)[k4];
    reinterpret_cast<float4 *>(h)[i] = make_float4(fmaf(a0, w01.x, fmaf(a1, w01.y, bb.x)), fmaf(a0, w01.z, fmaf(a1, w01.w, bb.y)),
                                                   fmaf(a0, w23.x, fmaf(a1, w23.y, bb.z)), fmaf(a0, w23.z, fmaf(a1, w23.w, bb.w)));
  }
}

// g_a[n,j] = sum_k g_h[n,k] W[k,j]; partial rows: w_partials[wave, (512, 2)] (layout of W), b_partials[wave, 512]
__global__ __launch_bounds__(256) void k_dec_fc_bwd(const float *__restrict__ g_h, const float *__restrict__ a,
                                                    const float *__restrict__ W, float *__restrict__ g_a,
                                                    float *__restrict__ w_partials, float *__restrict__ b_partials, int64_t N) {
  const int lane = threadIdx.x & 63;
  const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
  // lane's columns: k = 4 lane + {0..3} and 256 + 4 lane + {0..3}; W[k][0..1] pairs
  float4 w[2][2], aw[2][2], ab[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    w[h][0] = reinterpret_cast<const float4 *>(W)[2 * (h * 64 + lane)];        // (W[k][0], W[k][1], W[k+1][0], W[k+1][1])
    w[h][1] = reinterpret_cast<const float4 *>(W)[2 * (h * 64 + lane) + 1];    // (W[k+2][0], W[k+2][1], W[k+3][0], W[k+3][1])
    aw[h][0] = aw[h][1] = ab[h] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int64_t n = gw; n < N; n += nw) {
    const float4 g0 = reinterpret_cast<const float4 *>(g_h + n * HD_F)[lane], g1 = reinterpret_cast<const float4 *>(g_h + n * HD_F)[64 + lane];
    const float a0 = a[n * HD_A], a1 = a[n * HD_A + 1];
    float d0 = (g0.x * w[0][0].x + g0.y * w[0][0].z) + (g0.z * w[0][1].x + g0.w * w[0][1].z) +
               (g1.x * w[1][0].x + g1.y * w[1][0].z) + (g1.z * w[1][1].x + g1.w * w[1][1].z);
    float d1 = (g0.x * w[0][0].y + g0.y * w[0][0].w) + (g0.z * w[0][1].y + g0.w * w[0][1].w) +
               (g1.x * w[1][0].y + g1.y * w[1][0].w) + (g1.z * w[1][1].y + g1.w * w[1][1].w);
    d0 = hd_wave_sum(d0);
    d1 = hd_wave_sum(d1);
    if (lane == 0) { g_a[n * HD_A] = d0; g_a[n * HD_A + 1] = d1; }
    aw[0][0] = make_float4(fmaf(g0.x, a0, aw[0][0].x), fmaf(g0.x, a1, aw[0][0].y), fmaf(g0.y, a0, aw[0][0].z), fmaf(g0.y, a1, aw[0][0].w));
    aw[0][1] = make_float4(fmaf(g0.z, a0, aw[0][1].x), fmaf(g0.z, a1, aw[0][1].y), fmaf(g0.w, a0, aw[0][1].z), fmaf(g0.w, a1, aw[0][1].w));
    aw[1][0] = make_float4(fmaf(g1.x, a0, aw[1][0].x), fmaf(g1.x, a1, aw[1][0].y), fmaf(g1.y, a0, aw[1][0].z), fmaf(g1.y, a1, aw[1][0].w));
    aw[1][1] = make_float4(fmaf(g1.z, a0, aw[1][1].x), fmaf(g1.z, a1, aw[1][1].y), fmaf(g1.w, a0, aw[1][1].z), fmaf(g1.w, a1, aw[1][1].w));
    ab[0].x += g0.x; ab[0].y += g0.y; ab[0].z += g0.z; ab[0].w += g0.w;
    ab[1].x += g1.x; ab[1].y += g1.y; ab[1].z += g1.z; ab[1].w += g1.w;
  }
  float4 *wp = reinterpret_cast<float4 *>(w_partials + gw * (HD_F * HD_A));
  float4 *bp = reinterpret_cast<float4 *>(b_partials + gw * HD_F);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    wp[2 * (h * 64 + lane)] = aw[h][0];
    wp[2 * (h * 64 + lane) + 1] = aw[h][1];
    bp[h * 64 + lane] = ab[h];
  }
}

// reg[n] = log N(a;0,1) - log N(a;mu,var) summed over the latent dimensions (the 1/2 log 2 pi terms cancel)
__global__ __launch_bounds__(256) void k_latent_reg_fwd(const float *__restrict__ a, const float *__restrict__ mu,
                                                        const float *__restrict__ var, float *__restrict__ reg, int64_t N, int A) {
  for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
    float s = 0.f;
    for (int j = 0; j < A; ++j) {
      const float x = a[n * A + j], d = x - mu[n * A + j], v = var[n * A + j];
      s += (-0.5f * x * x) - (-0.5f * logf(v) - d * d / (2.f * v));
    }
    reg[n] = s;
  }
}
__global__ __launch_bounds__(256) void k_latent_reg_bwd(const float *__restrict__ a, const float *__restrict__ mu,
                                                        const float *__restrict__ var, const float *__restrict__ g,
                                                        float *__restrict__ g_a, float *__restrict__ g_mu, float *__restrict__ g_var,
                                                        int64_t N, int A) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N * A; i += (int64_t)gridDim.x * 256) {
    const float x = a[i], d = x - mu[i], v = var[i], gg = g[i / A];
    g_a[i] = gg * (-x + d / v);
    g_mu[i] = gg * (-d / v);
    g_var[i] = gg * (0.5f / v - d * d / (2.f * v * v));
  }
}

// The scalar head of the objective (reference kvae/vae/losses.py:45-69 and kvae/model/model.py:214-232) in ONE launch:
//   recon = sum(lpx * mk) / denom, reg = sum(regf * mk) / denom, denom = max(sum mk, 1)
//   vae_elbo = scale * recon + beta * reg;  elbo_total = vae_w * vae_elbo + kf_w * elbo_kf;  loss = -elbo_total
// out[6] = (loss, elbo_total, elbo_kf, vae_elbo, recon, reg); coef[3] = d loss / d lpx, d loss / d regf per observed frame, kf_w.
// w_dev (may be NULL) = device scalars (vae_w, kf_w) that override the by-value weights: the reference's phases move kf_weight
// between epochs (train.py:246-260) and a captured step has to follow.
// As torch ops this is ~25 dependent launches of a few microseconds each (and ~17 more in the backward), all on the
// critical path between the decoder's forward and backward.
__global__ __launch_bounds__(1024) void k_loss_head_fwd(const float *__restrict__ lpx, const float *__restrict__ regf,
                                                        const float *__restrict__ mask, const float *__restrict__ elbo_kf,
                                                        const float *__restrict__ beta, float scale, float vae_w, float kf_w,
                                                        const float *__restrict__ w_dev, float *__restrict__ out,
                                                        float *__restrict__ coef, int64_t n) {
  // one block of 1024 threads, dwordx4 loads: a handful of memory round trips in total (a narrower block turns the
  // reduction into a 50-deep chain of dependent global loads, slower than the launches it replaces)
  __shared__ float red[3][1024];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int64_t n4 = ((n & 3) == 0 && ((((uintptr_t)lpx | (uintptr_t)regf | (uintptr_t)mask) & 15) == 0)) ? n / 4 : 0;
  // mask == NULL means "every frame observed".  Loads are always issued from a VALID address (lpx stands in) and the value
  // selected afterwards: hipcc may hoist a load out of a short `mask ? mask[i] : 1` block (see mask_addr() in lgssm_fwd.h).
  const float *msrc = mask ? mask : lpx;
  for (int64_t i = threadIdx.x; i < n4; i += 1024) {
    const float4 a = reinterpret_cast<const float4 *>(lpx)[i], b = reinterpret_cast<const float4 *>(regf)[i];
    const float4 mr = reinterpret_cast<const float4 *>(msrc)[i];
    const float4 m = mask ? mr : make_float4(1.f, 1.f, 1.f, 1.f);
    s0 += (a.x * m.x + a.y * m.y) + (a.z * m.z + a.w * m.w);
    s1 += (b.x * m.x + b.y * m.y) + (b.z * m.z + b.w * m.w);
    s2 += (m.x + m.y) + (m.z + m.w);
  }
  for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 1024) {
    const float mraw = msrc[i];
    const float mk = mask ? mraw : 1.f;
    s0 = fmaf(lpx[i], mk, s0);
    s1 = fmaf(regf[i], mk, s1);
    s2 += mk;
  }
  red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
      red[2][threadIdx.x] += red[2][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float denom = fmaxf(red[2][0], 1.f), b = beta[0], kf = elbo_kf[0];
    if (w_dev) vae_w = w_dev[0], kf_w = w_dev[1];
    const float recon = red[0][0] / denom, reg = red[1][0] / denom;
    const float vae = scale * recon + b * reg, tot = vae_w * vae + kf_w * kf;
    out[0] = -tot; out[1] = tot; out[2] = kf; out[3] = vae; out[4] = recon; out[5] = reg;
    coef[0] = -vae_w * scale / denom;
    coef[1] = -vae_w * b / denom;
    coef[2] = kf_w;
  }
}
// g_lpx = g * coef[0] * mk, g_regf = g * coef[1] * mk, g_kf = -coef[2] * g   (g = upstream gradient of the loss)
__global__ __launch_bounds__(256) void k_loss_head_bwd(const float *__restrict__ g, const float *__restrict__ coef,
                                                       const float *__restrict__ mask, float *__restrict__ g_lpx,
                                                       float *__restrict__ g_regf, float *__restrict__ g_kf, int64_t n) {
  const float gg = g[0], c0 = gg * coef[0], c1 = gg * coef[1];
  const float *msrc = mask ? mask : g_lpx;   // always a valid address, value selected afterwards (see k_loss_head_fwd)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float mraw = msrc[i];
    const float mk = mask ? mraw : 1.f;
    g_lpx[i] = c0 * mk;
    g_regf[i] = c1 * mk;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) g_kf[0] = -coef[2] * gg;
}

}  // namespace kvae
