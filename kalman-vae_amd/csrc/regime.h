// regime.h — the sequential Gumbel-softmax regime chain of the switching dynamics:
//   y_0 = gs(init_logits, g_0);  l_t = y_{t-1}^T logits[t];  y_t = gs(l_t, g_t)
//   log q_t = sum_j y_t[j] log_softmax(l_t)[j];  log p_t = sum_j y_t[j] log(clamp(y_{t-1}^T P, 1e-8))[j]
// (reference: switch_dyn_param.py:52-79, F.gumbel_softmax with tau, hard = straight-through one-hot in eval).
// The reference runs this as T-1 Python iterations of ~10 aten ops each (plus their autograd mirror); here it is
// ONE launch forward and ONE launch backward (BPTT), one wavefront per sequence, the K x K tiles of a step in LDS.
#pragma once
#include "lgssm_vm.h"

#define KVAE_REGIME_MAX_K 16

namespace kvae {

struct RegimeLds {
  float Lt[KVAE_REGIME_MAX_K * KVAE_REGIME_MAX_K], P[KVAE_REGIME_MAX_K * KVAE_REGIME_MAX_K];
  float y[KVAE_REGIME_MAX_K], yp[KVAE_REGIME_MAX_K], l[KVAE_REGIME_MAX_K], g[KVAE_REGIME_MAX_K];
  float tp[KVAE_REGIME_MAX_K], ys[KVAE_REGIME_MAX_K], gy[KVAE_REGIME_MAX_K], gl[KVAE_REGIME_MAX_K], gtp[KVAE_REGIME_MAX_K];
  float gyin[KVAE_REGIME_MAX_K];   // upstream g_y of the current step
};

// softmax / log_softmax pieces of a K-vector held in LDS, computed redundantly by the calling lane
KV_DEV void lse_of(const float *v, int K, float scale, const float *add, float *mx_out, float *lse_out) {
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) mx = fmaxf(mx, (v[k] + (add ? add[k] : 0.f)) * scale);
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += expf((v[k] + (add ? add[k] : 0.f)) * scale - mx);
  *mx_out = mx;
  *lse_out = logf(s);
}

// Gumbel-softmax sample of lane j: soft value and (if hard) the straight-through one-hot value
KV_DEV float gs_sample(const float *l, const float *g, int K, float inv_tau, int hard, int j, float *soft_out) {
  float mx, lse;
  lse_of(l, K, inv_tau, g, &mx, &lse);
  const float soft = expf((l[j] + g[j]) * inv_tau - mx) / expf(lse);
  *soft_out = soft;
  if (!hard) return soft;
  int arg = 0;
  float best = -INFINITY;
  for (int k = 0; k < K; ++k) {
    const float sk = expf((l[k] + g[k]) * inv_tau - mx) / expf(lse);
    if (sk > best) { best = sk; arg = k; }
  }
  return ((j == arg ? 1.0f : 0.0f) - soft) + soft;
}

KV_DEV void regime_fwd_body(const float *logits, const float *init_logits, const float *gumbel, const float *Pm,
                            float *y_seq, float *log_q, float *log_p, int b, int T, int K, float tau, int hard,
                            RegimeLds &L) {
  const float inv_tau = 1.0f / tau;
  const int KK = K * K;
  KV_PAR(e, KK) { L.P[e] = Pm[e]; }
  KV_PAR(j, K) { L.l[j] = init_logits[(int64_t)b * K + j]; L.g[j] = gumbel[((int64_t)b * T) * K + j]; L.yp[j] = 0.f; }
  // the inputs of step t+1 do not depend on step t: they are fetched into registers one step ahead, otherwise every
  // step of this latency-bound chain starts with a full global-memory round trip
  Prefetch<KVAE_REGIME_MAX_K * KVAE_REGIME_MAX_K> pf_l;
  Prefetch<KVAE_REGIME_MAX_K> pf_g;
  if (T > 1) {
    pf_l.issue(logits + ((int64_t)b * T + 1) * KK, KK);
    pf_g.issue(gumbel + ((int64_t)b * T + 1) * K, K);
  }
  KV_SYNC();
  for (int t = 0; t < T; ++t) {
    const int64_t q = (int64_t)b * T + t;
    if (t > 0) {
      pf_l.commit(L.Lt, KK);
      pf_g.commit(L.g, K);
      KV_SYNC();
      if (t + 1 < T) {
        pf_l.issue(logits + (q + 1) * KK, KK);
        pf_g.issue(gumbel + (q + 1) * K, K);
      }
      KV_PAR(j, K) {  // l_t = y_{t-1}^T logits[t] ; tp = y_{t-1}^T P
        float acc = 0.f, acp = 0.f;
        for (int i = 0; i < K; ++i) {
          acc = fmaf(L.yp[i], L.Lt[i * K + j], acc);
          acp = fmaf(L.yp[i], L.P[i * K + j], acp);
        }
        L.l[j] = acc;
        L.tp[j] = acp;
      }
      KV_SYNC();
    }
    KV_PAR(j, K) {
      float soft;
      const float yj = gs_sample(L.l, L.g, K, inv_tau, hard, j, &soft);
      L.y[j] = yj;
      y_seq[q * K + j] = yj;
    }
    KV_SYNC();
    KV_LANE0 {
      float mx, lse, lq = 0.f, lp = 0.f;
      lse_of(L.l, K, 1.0f, nullptr, &mx, &lse);
      for (int j = 0; j < K; ++j) {
        lq = fmaf(L.y[j], (L.l[j] - mx) - lse, lq);
        lp = fmaf(L.y[j], t > 0 ? logf(fmaxf(L.tp[j], 1e-8f)) : logf(1.0f / (float)K), lp);
      }
      log_q[q] = lq;
      log_p[q] = lp;
    }
    KV_PAR(j, K) { L.yp[j] = L.y[j]; }   // yp is not read in this phase
    KV_SYNC();
  }
}

// BPTT. Upstream: g_y [B,T,K], g_lq [B,T], g_lp [B,T]. Outputs: g_logits [B,T,K,K] (slice t = 0 zeroed), g_init [B,K].
KV_DEV void regime_bwd_body(const float *logits, const float *init_logits, const float *gumbel, const float *Pm,
                            const float *y_seq, const float *g_y, const float *g_lq, const float *g_lp, float *g_logits,
                            float *g_init, int b, int T, int K, float tau, RegimeLds &L) {
  const float inv_tau = 1.0f / tau;
  const int KK = K * K;
  KV_PAR(e, KK) { L.P[e] = Pm[e]; }
  KV_PAR(j, K) { L.gy[j] = 0.f; }
  // one-step-ahead register prefetch of everything a step reads from global memory (see the forward body)
  Prefetch<KVAE_REGIME_MAX_K * KVAE_REGIME_MAX_K> pf_l;
  Prefetch<KVAE_REGIME_MAX_K> pf_y, pf_yp, pf_g, pf_gy;
  float glq_n = 0.f, glp_n = 0.f;
  auto issue = [&](int t) {
    const int64_t q = (int64_t)b * T + t;
    pf_y.issue(y_seq + q * K, K);
    pf_g.issue(gumbel + q * K, K);
    pf_gy.issue(g_y + q * K, K);
    if (t > 0) {
      pf_yp.issue(y_seq + (q - 1) * K, K);
      pf_l.issue(logits + q * KK, KK);
    }
    glq_n = g_lq[q];
    glp_n = g_lp[q];
  };
  issue(T - 1);
  KV_SYNC();
  for (int t = T - 1; t >= 0; --t) {
    const int64_t q = (int64_t)b * T + t;
    pf_y.commit(L.y, K);
    pf_g.commit(L.g, K);
    pf_gy.commit(L.gyin, K);
    if (t > 0) {
      pf_yp.commit(L.yp, K);
      pf_l.commit(L.Lt, KK);
    } else {
      KV_PAR(j, K) { L.yp[j] = 0.f; L.l[j] = init_logits[(int64_t)b * K + j]; }
    }
    const float glq = glq_n, glp = glp_n;
    KV_SYNC();
    if (t > 0) issue(t - 1);
    if (t > 0) {
      KV_PAR(j, K) {
        float acc = 0.f, acp = 0.f;
        for (int i = 0; i < K; ++i) {
          acc = fmaf(L.yp[i], L.Lt[i * K + j], acc);
          acp = fmaf(L.yp[i], L.P[i * K + j], acp);
        }
        L.l[j] = acc;
        L.tp[j] = acp;
      }
      KV_SYNC();
    }
    // total adjoint of y_t, and the soft sample (needed for the softmax Jacobian)
    KV_PAR(j, K) {
      float mx, lse, mx1, lse1;
      lse_of(L.l, K, inv_tau, L.g, &mx, &lse);
      lse_of(L.l, K, 1.0f, nullptr, &mx1, &lse1);
      L.ys[j] = expf((L.l[j] + L.g[j]) * inv_tau - mx) / expf(lse);
      const float lsm = (L.l[j] - mx1) - lse1;
      const float lpj = t > 0 ? logf(fmaxf(L.tp[j], 1e-8f)) : logf(1.0f / (float)K);
      L.gy[j] = L.gy[j] + L.gyin[j] + glq * lsm + glp * lpj;   // same lane reads and writes gy[j]
    }
    KV_SYNC();
    KV_PAR(j, K) {
      float dotv = 0.f, sy = 0.f, mx1, lse1;
      for (int k = 0; k < K; ++k) { dotv = fmaf(L.gy[k], L.ys[k], dotv); sy += L.y[k]; }
      lse_of(L.l, K, 1.0f, nullptr, &mx1, &lse1);
      const float sm1 = expf((L.l[j] - mx1) - lse1);
      // through the (soft) sample, and the direct dependence of log q_t on l_t
      L.gl[j] = L.ys[j] * (L.gy[j] - dotv) * inv_tau + glq * (L.y[j] - sm1 * sy);
      L.gtp[j] = (t > 0 && L.tp[j] >= 1e-8f) ? glp * L.y[j] / L.tp[j] : 0.f;
    }
    KV_SYNC();
    if (t > 0) {
      KV_PAR(e, KK) {
        const int i = e / K, j = e - i * K;
        g_logits[q * KK + e] = L.yp[i] * L.gl[j];
      }
      KV_PAR(i, K) {  // adjoint of y_{t-1}
        float acc = 0.f;
        for (int j = 0; j < K; ++j) {
          acc = fmaf(L.Lt[i * K + j], L.gl[j], acc);
          acc = fmaf(L.P[i * K + j], L.gtp[j], acc);
        }
        L.ys[i] = acc;   // staging (ys was last read in the previous phase)
      }
    } else {
      KV_PAR(e, KK) { g_logits[q * KK + e] = 0.f; }
      KV_PAR(j, K) { g_init[(int64_t)b * K + j] = L.gl[j]; }
    }
    KV_SYNC();
    if (t > 0) { KV_PAR(i, K) { L.gy[i] = L.ys[i]; } }
    KV_SYNC();
  }
}

}  // namespace kvae
