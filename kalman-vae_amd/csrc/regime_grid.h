// regime_grid.h — the Gumbel-softmax regime chain of the switching dynamics (regime.h: same equations, reference
// switch_dyn_param.py:52-79) for K <= 8 with ONE WAVEFRONT PER SEQUENCE laid out as an 8 x 8 grid of lanes, registers only.
//
// Why: at BASELINE configs[3] (K = 7, T = 100, 32 sequences) the chain is a third of the training step.  The thread-per-sequence
// kernels (kvae_lgssm_tpp.hip) walk a step's K x K logits with ~50 dependent scalar loads and ~200 serial flops per thread
// (235 us forward / 325 us backward there: 2.4 / 3.3 us per step on 32 busy lanes of the whole chip); the LDS version (regime.h)
// pays a round trip and a barrier per phase.  Here a step is a handful of DPP reductions and one cross-lane permute:
//
//   lane = 8 g + k.  A vector indexed by a regime lives either "along the lanes" (A-layout: element k on lane k of EVERY group)
//   or "on the groups" (G-layout: element g on all eight lanes of group g).
//   forward:  lane (g, k) holds logits[t][i = k][j = g]:   l_t[g] = sum_k y_{t-1}[k] logits[k][g] is an 8-lane DPP all-reduce
//             (G-layout); one ds_bpermute turns it into the A-layout in which softmax / argmax / the two log terms are again
//             8-lane reductions - and in which y_t is exactly the operand of the next step.  Every group computes the sample
//             redundantly; nothing is ever broadcast from a single lane.
//   backward: the same recomputation, the adjoint of l_t in A-layout; the outer product y_{t-1}[i] gl[j] is written from the
//             natural assignment lane (g, k) <-> element [i = g][j = k] (y_{t-1}[g] straight from memory, gl[k] on the lane),
//             and the adjoint of y_{t-1}, sum_j logits[i][j] gl[j] + P[i][j] gtp[j], is an 8-lane reduction in that same
//             assignment (G-layout) plus one permute.
//   The next step's operands are fetched one step ahead; loads are unconditional from clamped addresses (see mask_addr()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kvae {
namespace rgrid {

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// all-reduce over the 8 lanes of a group: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
__device__ __forceinline__ float oct_sum(float x) {
  x += dpp<0xB1>(x);
  x += dpp<0x4E>(x);
  x += dpp<0x141>(x);
  return x;
}
__device__ __forceinline__ float oct_max(float x) {
  x = fmaxf(x, dpp<0xB1>(x));
  x = fmaxf(x, dpp<0x4E>(x));
  x = fmaxf(x, dpp<0x141>(x));
  return x;
}
__device__ __forceinline__ float oct_min(float x) {
  x = fminf(x, dpp<0xB1>(x));
  x = fminf(x, dpp<0x4E>(x));
  x = fminf(x, dpp<0x141>(x));
  return x;
}
// G-layout -> A-layout: lane (g, k) takes the value group k holds
__device__ __forceinline__ float g2a(float vG, int k) { return __shfl(vG, k << 3, 64); }

struct Soft {
  float soft, mx1, lse1;   // softmax((l + g) / tau)[k];  max and log-sum-exp of l itself
};
__device__ __forceinline__ Soft soft_of(float l, float g, float inv_tau, bool vk) {
  const float v = vk ? (l + g) * inv_tau : -INFINITY;
  const float mx = oct_max(v);
  const float e = vk ? __expf(v - mx) : 0.0f;
  const float s = oct_sum(e);
  Soft r;
  r.soft = e / s;
  r.mx1 = oct_max(vk ? l : -INFINITY);
  r.lse1 = __logf(oct_sum(vk ? __expf(l - r.mx1) : 0.0f));
  return r;
}

__device__ __forceinline__ void regime_fwd(const float *__restrict__ logits, const float *__restrict__ init_logits,
                                           const float *__restrict__ gumbel, const float *__restrict__ Pm,
                                           float *__restrict__ y_seq, float *__restrict__ log_q, float *__restrict__ log_p, int b,
                                           int T, int K, float tau, int hard) {
  const int lane = threadIdx.x & 63, k = lane & 7, g = lane >> 3;
  const bool vk = k < K, ve = vk && g < K;
  const int KK = K * K, e1 = ve ? k * K + g : 0, kc = vk ? k : 0;
  const float inv_tau = 1.0f / tau, log_uniform = __logf(1.0f / (float)K);
  const float P1r = Pm[e1], P1 = ve ? P1r : 0.0f;
  const int64_t q0 = (int64_t)b * T;
  const float l0r = init_logits[(int64_t)b * K + kc];
  float l = vk ? l0r : 0.0f, tp = 1.0f, y = 0.0f;
  float gum = gumbel[q0 * K + kc];
  // operands of step 1, in flight during step 0
  int64_t qn = q0 + (T > 1 ? 1 : 0);
  float Ln = logits[qn * KK + e1], gn = gumbel[qn * K + kc];
  for (int t = 0; t < T; ++t) {
    const int64_t q = q0 + t;
    if (t > 0) {
      const float Lt = ve ? Ln : 0.0f;
      gum = gn;
      qn = q + (t + 1 < T ? 1 : 0);
      Ln = logits[qn * KK + e1];
      gn = gumbel[qn * K + kc];
      const float lG = oct_sum(y * Lt), tG = oct_sum(y * P1);   // y is y_{t-1} here
      l = g2a(lG, k);
      tp = g2a(tG, k);
    }
    const Soft s = soft_of(l, gum, inv_tau, vk);
    float yt = s.soft;
    if (hard) {   // straight-through one-hot of the FIRST maximum (F.gumbel_softmax(hard=True): argmax)
      const float best = oct_max(vk ? s.soft : -INFINITY);
      const float arg = oct_min((vk && s.soft == best) ? (float)k : 99.0f);
      yt = (((float)k == arg ? 1.0f : 0.0f) - s.soft) + s.soft;
    }
    yt = vk ? yt : 0.0f;
    if (lane < K) y_seq[q * K + lane] = yt;
    const float lsm = (l - s.mx1) - s.lse1;
    const float lpj = t > 0 ? __logf(fmaxf(tp, 1e-8f)) : log_uniform;
    const float lq = oct_sum(vk ? yt * lsm : 0.0f), lp = oct_sum(vk ? yt * lpj : 0.0f);
    if (lane == 0) {
      log_q[q] = lq;
      log_p[q] = lp;
    }
    y = yt;
  }
}

// BPTT.  Upstream: g_y [B,T,K], g_lq [B,T], g_lp [B,T].  Outputs: g_logits [B,T,K,K] (slice t = 0 zeroed), g_init [B,K].
__device__ __forceinline__ void regime_bwd(const float *__restrict__ logits, const float *__restrict__ init_logits,
                                           const float *__restrict__ gumbel, const float *__restrict__ Pm,
                                           const float *__restrict__ y_seq, const float *__restrict__ g_y,
                                           const float *__restrict__ g_lq, const float *__restrict__ g_lp,
                                           float *__restrict__ g_logits, float *__restrict__ g_init, int b, int T, int K,
                                           float tau) {
  const int lane = threadIdx.x & 63, k = lane & 7, g = lane >> 3;
  const bool vk = k < K, vg = g < K, ve = vk && vg;
  const int KK = K * K, e1 = ve ? k * K + g : 0, e2 = ve ? g * K + k : 0, kc = vk ? k : 0, gc = vg ? g : 0;
  const float inv_tau = 1.0f / tau, log_uniform = __logf(1.0f / (float)K);
  const float P1r = Pm[e1], P2r = Pm[e2];
  const float P1 = ve ? P1r : 0.0f, P2 = ve ? P2r : 0.0f;
  const int64_t q0 = (int64_t)b * T;
  const float l0r = init_logits[(int64_t)b * K + kc];
  float carry = 0.0f;                                            // adjoint of y_t handed down from step t + 1 (A-layout)
  // operands of step T - 1
  int64_t q = q0 + T - 1, qp = q - (T > 1 ? 1 : 0);
  float yN = y_seq[q * K + kc], gumN = gumbel[q * K + kc], gyN = g_y[q * K + kc], ypN = y_seq[qp * K + kc],
        ypGN = y_seq[qp * K + gc], L1N = logits[q * KK + e1], L2N = logits[q * KK + e2], glqN = g_lq[q], glpN = g_lp[q];
  for (int t = T - 1; t >= 0; --t) {
    q = q0 + t;
    const float y = vk ? yN : 0.0f, gum = gumN, gyin = vk ? gyN : 0.0f, glq = glqN, glp = glpN;
    const float yp = (vk && t > 0) ? ypN : 0.0f, ypG = (vg && t > 0) ? ypGN : 0.0f;
    const float L1 = ve ? L1N : 0.0f, L2 = ve ? L2N : 0.0f;
    if (t > 0) {   // operands of step t - 1
      const int64_t qn = q - 1, qnp = qn - (t > 1 ? 1 : 0);
      yN = y_seq[qn * K + kc], gumN = gumbel[qn * K + kc], gyN = g_y[qn * K + kc], ypN = y_seq[qnp * K + kc];
      ypGN = y_seq[qnp * K + gc], L1N = logits[qn * KK + e1], L2N = logits[qn * KK + e2], glqN = g_lq[qn], glpN = g_lp[qn];
    }
    float l = vk ? l0r : 0.0f, tp = 1.0f;
    if (t > 0) {
      const float lG = oct_sum(yp * L1), tG = oct_sum(yp * P1);
      l = g2a(lG, k);
      tp = g2a(tG, k);
    }
    const Soft s = soft_of(l, gum, inv_tau, vk);
    const float lsm = (l - s.mx1) - s.lse1;
    const float lpj = t > 0 ? __logf(fmaxf(tp, 1e-8f)) : log_uniform;
    const float gy = vk ? carry + gyin + glq * lsm + glp * lpj : 0.0f;   // total adjoint of y_t
    const float dotv = oct_sum(gy * s.soft), sy = oct_sum(y);
    const float sm1 = __expf(lsm);
    // through the (soft) sample, and the direct dependence of log q_t on l_t
    const float gl = vk ? s.soft * (gy - dotv) * inv_tau + glq * (y - sm1 * sy) : 0.0f;
    const float gtp = (vk && t > 0 && tp >= 1e-8f) ? glp * y / tp : 0.0f;
    if (t > 0) {
      if (ve) g_logits[q * KK + e2] = ypG * gl;                    // [i = g][j = k] = y_{t-1}[i] gl[j]
      const float cG = oct_sum(L2 * gl + P2 * gtp);                // adjoint of y_{t-1}[g]
      carry = g2a(cG, k);
    } else {
      if (ve) g_logits[q * KK + e2] = 0.0f;
      if (lane < K) g_init[(int64_t)b * K + lane] = gl;
    }
  }
}

}  // namespace rgrid
}  // namespace kvae
